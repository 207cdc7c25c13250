#!/bin/bash
mkdir -p gpurun_out/r04c28
timeout 100 python bench.py > gpurun_out/r04c28/bench_default.json 2> gpurun_out/r04c28/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c28/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], d['dtype'], d['step_frac_mfma'], d['launches_per_step'], d['non_mlp_us'], d['mlp_kernels_us_per_step'], '| dW', r['avg_launch_us'], r['frac'], r['traffic'], '| f8', d.get('f8_dw_ms_per_step'), '| ns', d['roofline_north_star']['avg_launch_us'], '| cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
