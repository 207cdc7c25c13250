# r05 call 9: PMC issue accounts of the step's MLP kernels (bf16), then the current default bench line with its in-run traffic passes
set -u
OUT=gpurun_out/r05_09; mkdir -p $OUT
tools/pmc_issue.sh r05_09/pmc bf16 > $OUT/pmc_issue.log 2>&1
cat gpurun_out/r05_09/pmc/pmc_issue.txt | cut -c1-260 | head -60
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_09/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: d.get(k) for k in ('launches_per_step','non_mlp_us','mlp_kernels_us_per_step','step_frac_mfma','f8_dw_ms_per_step','eval_ms_per_image')})
for k in d['roofline_kernels']: print(k['kernel'][:60], k['avg_launch_us'], k['frac_mfma'], k['frac_hbm'], k['traffic'])
print(d['roofline']); print(d.get('traffic_note')); print(d['cpu_baseline'])
PY
cat $OUT/bench_default.time
