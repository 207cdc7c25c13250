# r06 call 10: store-burst variants of the saving forward (NERFHIP_SAVE_BURST) and the chain (NERFHIP_CHAIN_BURST): parity of the variants, then ABAB in the step
set -u
OUT=gpurun_out/r06_10; mkdir -p $OUT
V=nerf_pl_amd/variants
for L in libnerfhip_b4.so libnerfhip_b8.so; do
  ( NERFHIP_LIB_PATH=$PWD/$V/$L timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_render_fused.py tests/test_gpu_fused_step.py -q -m gpu -x 2>&1 | tail -2 ) | sed "s/^/$L: /" | tee -a $OUT/pytest_variants.txt
done
for rep in 1 2 3; do
  for L in "" libnerfhip_s2c2.so libnerfhip_b4.so libnerfhip_b8.so libnerfhip_s1c16.so; do
    if [ -n "$L" ]; then export NERFHIP_LIB_PATH=$PWD/$V/$L; else unset NERFHIP_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ.get('NERFHIP_LIB_PATH','HEAD').split('/')[-1], d['ms_per_step'], [(k['kernel'][:18], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels'][:3]])"
  done
done | tee $OUT/burst_ab.txt
