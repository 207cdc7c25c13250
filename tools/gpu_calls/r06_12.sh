# r06 call 12: fold kernel rewritten (one memory round trip per workgroup): gradient tests, then the dW split-plan cost model re-swept on the new job mix
set -u
OUT=gpurun_out/r06_12; mkdir -p $OUT
( timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_step.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" ) | tee $OUT/pytest_subset.txt
for rep in 1 2 3; do
  for AB in "150 45" "100 50" "200 40" "50 55" "250 35" "150 40"; do
    set -- $AB
    export NERFHIP_DW_COST_A=$1 NERFHIP_DW_COST_B=$2; python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>$OUT/bench.err | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cost', os.environ['NERFHIP_DW_COST_A'], os.environ['NERFHIP_DW_COST_B'], d['ms_per_step'], d['literal_contract']['ms_per_step'], [(k['kernel'][:22], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels']])"
  done
done | tee $OUT/dw_plan_ab.txt
