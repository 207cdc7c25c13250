# r06 call 3: the dW kernel with the 2x4 wave split + dot2 bias sums + register-major partial slabs: correctness (backward tests of the
# three arithmetic modes), then kernel-alone and whole-step A/B against the build before the change and the single-feature variants
set -u
OUT=gpurun_out/r06_03; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_fused_step.py tests/test_gpu_layered.py -q -m gpu -x 2>&1 | tail -8 ) | tee $OUT/pytest_bwd.txt
V=nerf_pl_amd/variants
for rep in 1 2; do
  for L in libnerfhip_r06base.so "" libnerfhip_dw_nosplit.so libnerfhip_dw_nodot2.so libnerfhip_dw_rd3.so libnerfhip_dw_rd5.so; do
    if [ -n "$L" ]; then export NERFHIP_LIB_PATH=$PWD/$V/$L; else unset NERFHIP_LIB_PATH; fi
    python tools/kbench.py --merged --reps 30 2>/dev/null | tail -1
  done
done | tee $OUT/kbench_dw_ab.txt
for rep in 1 2; do
  for L in libnerfhip_r06base.so "" libnerfhip_dw_nosplit.so; do
    if [ -n "$L" ]; then export NERFHIP_LIB_PATH=$PWD/$V/$L; else unset NERFHIP_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ.get('NERFHIP_LIB_PATH','HEAD').split('/')[-1], d['ms_per_step'], d['literal_contract']['ms_per_step'], [(k['kernel'][:22], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels']])"
  done
done | tee $OUT/bench_ab.txt
