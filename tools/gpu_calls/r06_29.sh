# r06 call 29: Adam launch granularity (NERFHIP_ADAM_R = float4 per thread: 4 = 292 workgroups (rounds 2-6) | 2 | 1 = 1164 workgroups):
# optimizer tests, then same-library ABAB
set -u
OUT=gpurun_out/r06_29; mkdir -p $OUT
for R in 1 2; do ( NERFHIP_ADAM_R=$R timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_step.py -q -m gpu -k "adam or Adam or optim or step" 2>&1 | grep -E "passed|failed|FAILED|Error" | cut -c1-300 ) 2>&1 | sed "s/^/R=$R /" | tee -a $OUT/pytest_adam.txt; done
for rep in 1 2 3 4; do
  for R in 4 2 1; do
    NERFHIP_ADAM_R=$R python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null | R=$R python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('adam R=%s' % os.environ['R'], 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], 'mlp kernels', d['mlp_kernels_us_per_step'], 'non-mlp', d['non_mlp_us'])"
  done
done | tee $OUT/adam_abab.txt
