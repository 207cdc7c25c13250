# r06 call 26: the bf16 saving forward stores its six input-encoding slabs ahead of the dir layer (from the LDS stash) instead of in the
# prologue (NERFHIP_ENC_SAVE_LATE, variants/libnerfhip_late.so) — tests under that library, then same-box ABAB against the in-tree library
set -u
OUT=gpurun_out/r06_26; mkdir -p $OUT
L=$PWD/nerf_pl_amd/variants/libnerfhip_late.so
( time NERFHIP_LIB_PATH=$L timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_fused_step.py tests/test_gpu_bf16.py tests/test_gpu_render_fused.py tests/test_gpu_layered.py tests/test_gpu_draws.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | cut -c1-400 ) 2>&1 | tee $OUT/pytest_subset_late.txt
for rep in 1 2 3 4; do
  for V in base late; do
    if [ $V = late ]; then export NERFHIP_LIB_PATH=$L; else unset NERFHIP_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null | V=$V python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-5s' % os.environ['V'], 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], [(k['kernel'][:20], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels']], 'non-mlp', d['non_mlp_us'])"
  done
done | tee $OUT/late_abab.txt
