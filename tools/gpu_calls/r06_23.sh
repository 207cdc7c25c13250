# r06 call 23: regenerated encodings — rays preloaded one iteration ahead by scalar loads, the slabs formed BEHIND the stage's MFMAs, the
# split plan left as it is (NERFHIP_DW_REGEN_PLAN=1: priced by the bytes still fetched): tests, then same-tree ABAB
set -u
OUT=gpurun_out/r06_23; mkdir -p $OUT
( time timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_step.py tests/test_gpu_render_fused.py -q -m gpu -s 2>&1 | grep -E "passed|failed|FAILED|Error|assert|regenerated vs saved" | cut -c1-400 ) 2>&1 | tee $OUT/pytest_subset.txt
for rep in 1 2 3; do
  for R in 0 1 2; do
    P=0; RR=$R; if [ $R = 2 ]; then P=1; RR=1; fi
    NERFHIP_DW_REGEN_PLAN=$P NERFHIP_REGEN_ENC=$RR python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null | R=$R python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({'0':'saved        ','1':'regen        ','2':'regen + plan '}[os.environ['R']], 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], [(k['kernel'][:20], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels']], 'non-mlp', d['non_mlp_us'])"
  done
done | tee $OUT/regen_abab.txt
