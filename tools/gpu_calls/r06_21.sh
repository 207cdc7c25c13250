# r06 call 21: input encodings regenerated in the dW launch instead of saved (nerfhip_render_args.regen_enc + nerfhip_mlp_bwd_multi_rays):
# the new tests, the gradient / step suites, then same-tree ABAB through the env switch (NERFHIP_REGEN_ENC=0 | 1)
set -u
OUT=gpurun_out/r06_21; mkdir -p $OUT
( time timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_step.py -q -m gpu -x -s -k "regenerated or without_saved or fused_step_equals_modular" 2>&1 | grep -E "passed|failed|FAILED|Error|assert|regenerated vs saved|fused vs modular" | cut -c1-400 ) 2>&1 | tee $OUT/pytest_regen.txt
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_fused_step.py tests/test_gpu_bf16.py tests/test_gpu_render_fused.py tests/test_gpu_layered.py tests/test_gpu_inference.py tests/test_gpu_draws.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | cut -c1-400 ) 2>&1 | tee $OUT/pytest_subset.txt
for rep in 1 2 3; do
  for R in 0 1; do
    NERFHIP_REGEN_ENC=$R python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null | R=$R python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('regen_enc=%s' % os.environ['R'], 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], [(k['kernel'][:20], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels']], 'non-mlp', d['non_mlp_us'])"
  done
done | tee $OUT/regen_abab.txt
