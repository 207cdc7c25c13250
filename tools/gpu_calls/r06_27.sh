# r06 call 27: persistent backward chain (bf16): one workgroup per CU runs every 256th block, no workgroup hand-over between the blocks
# of a CU.  Gradient / step suites, then same-library ABAB through NERFHIP_CHAIN_PERSIST=0 (one block per workgroup, rounds 1-6) | default
set -u
OUT=gpurun_out/r06_27; mkdir -p $OUT
( time timeout 2400 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_step.py tests/test_gpu_bf16.py tests/test_gpu_render_fused.py tests/test_gpu_layered.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | cut -c1-400 ) 2>&1 | tee $OUT/pytest_subset.txt
for rep in 1 2 3 4; do
  for V in 0 256 512; do
    NERFHIP_CHAIN_PERSIST=$V python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null | V=$V python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('persist=%-4s' % os.environ['V'], 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], [(k['kernel'][:20], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels']], 'non-mlp', d['non_mlp_us'])"
  done
done | tee $OUT/persist_abab.txt
