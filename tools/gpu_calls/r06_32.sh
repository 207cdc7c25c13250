# r06 call 32: sub-pass boundaries of the single-launch render kernels WITHOUT the store drain where no compositing tail follows
# (fine -> fine: barrier + lgkmcnt only; variants/libnerfhip_lsync.so): bit-equality tests under that library, then same-box ABAB
set -u
OUT=gpurun_out/r06_32; mkdir -p $OUT
L=$PWD/nerf_pl_amd/variants/libnerfhip_lsync.so
( time NERFHIP_LIB_PATH=$L timeout 2400 python -m pytest tests/test_gpu_render_fused.py tests/test_gpu_fused_step.py tests/test_gpu_training.py tests/test_gpu_inference.py tests/test_gpu_bf16.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | cut -c1-400 ) 2>&1 | tee $OUT/pytest_subset_lsync.txt
for rep in 1 2 3 4; do
  for V in base lsync; do
    if [ $V = lsync ]; then export NERFHIP_LIB_PATH=$L; else unset NERFHIP_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null | V=$V python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-5s' % os.environ['V'], 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], [(k['kernel'][:20], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels']], 'non-mlp', d['non_mlp_us'])"
  done
done | tee $OUT/lsync_abab.txt
for V in base lsync; do
  if [ $V = lsync ]; then export NERFHIP_LIB_PATH=$L; else unset NERFHIP_LIB_PATH; fi
  python bench.py --mode eval --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | V=$V python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-5s eval' % os.environ['V'], d.get('ms_per_step'), d.get('value'))"
done | tee $OUT/lsync_eval.txt
