# r06 call 31: regenerated encodings formed as HALF slabs (xyz by all 8 waves, dir by the dir job's idle waves 4..7), the depths read from
# LDS at the top of the iteration: the regen tests + gradient / step suites, then same-library ABAB (NERFHIP_REGEN_ENC=0 | 1)
set -u
OUT=gpurun_out/r06_31; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_step.py tests/test_gpu_render_fused.py tests/test_gpu_bf16.py -q -m gpu -s 2>&1 | grep -E "passed|failed|FAILED|Error|assert|regenerated vs saved" | cut -c1-400 ) 2>&1 | tee $OUT/pytest_subset.txt
for rep in 1 2 3 4 5; do
  for R in 0 1; do
    NERFHIP_REGEN_ENC=$R python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null | R=$R python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({'0':'saved','1':'regen'}[os.environ['R']], 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], [(k['kernel'][:20], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels']], 'non-mlp', d['non_mlp_us'])"
  done
done | tee $OUT/regen_abab.txt
