# r05 call 18: final bench.py (in-step kernel durations in the roofline) — the driver's command, the same under rocprofv3, smoke()
set -u
OUT=gpurun_out/r05_18; mkdir -p $OUT
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err ) 2> $OUT/bench_driver_cmd.time
tools/ktrace_step.sh r05_18/trace --gpus 1 > $OUT/kernel_by_grid.txt 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_18/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('cold_start_ms_per_step'), d['roofline'])
for k in d['roofline_kernels']: print(k['kernel'][:40], k['avg_launch_us'], k['in_step_launch_us'], k['frac_mfma_in_step'], k['frac_hbm_in_step'], k['traffic'])
print({k: d.get(k) for k in ('launches_per_step','non_mlp_us','mlp_kernels_us_per_step','step_frac_mfma','f8_dw_ms_per_step','eval_ms_per_image')}, d['cpu_baseline'])
PY
grep -E "dw_kernel<1>|render_kernel<1, 1>|chain_kernel<1, false> +524288" $OUT/kernel_by_grid.txt; cat $OUT/bench_driver_cmd.time
