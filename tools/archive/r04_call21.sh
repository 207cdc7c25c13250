#!/bin/bash
# e4m3 dW: split plan cost sweep (a + b x KiB per iteration) against equal iterations
OUT=gpurun_out/r04c21; mkdir -p $OUT
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc --dtype bf16_f8 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
}
for rep in 1 2; do
  run eqiter_$rep X=1
  run a650b22_$rep NERFHIP_DW_COST_A=650 NERFHIP_DW_COST_B=22
  run a900b15_$rep NERFHIP_DW_COST_A=900 NERFHIP_DW_COST_B=15
  run a1300b10_$rep NERFHIP_DW_COST_A=1300 NERFHIP_DW_COST_B=10
done
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=d.get('roofline_kernels') or []
    print(sys.argv[1].split('/')[-1].ljust(24), d['value'], d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')}, ' '.join('%s %.1f' % (k['kernel'].split('<')[0][4:]+('F' if 'fine pass' in k['kernel'] else 'C' if 'coarse pass' in k['kernel'] else ''), k['avg_launch_us']) for k in ks))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
