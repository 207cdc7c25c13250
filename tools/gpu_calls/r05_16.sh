# r05 call 16: final build — the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5) and the default one, the same under
# rocprofv3 --kernel-trace --stats, the other modes, the whole -m gpu suite
set -u
OUT=gpurun_out/r05_16; mkdir -p $OUT
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err ) 2> $OUT/bench_driver_cmd.time
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tools/ktrace_step.sh r05_16/trace > $OUT/kernel_by_grid.txt 2>&1
python bench.py --dtype bf16_f8 --no-cpu-baseline --no-extras > $OUT/bench_train_f8_dw.json 2>/dev/null
python bench.py --dtype fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_train_fp32.json 2>/dev/null
python bench.py --mode render --no-cpu-baseline > $OUT/bench_render.json 2>/dev/null
python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_eval.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --no-pmc --settle 0 --steps 20 --warmup 5 > $OUT/bench_settle0.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --force-dist --sync-in-graph 0 > $OUT/bench_rccl_world1_two_graphs.json 2>/dev/null
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('cold_start_ms_per_step', 'launches_per_step', 'non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')}, (d.get('roofline') or {}).get('avg_launch_us'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
tail -14 $OUT/kernel_by_grid.txt; cat $OUT/bench_driver_cmd.time
( time timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | grep -E "passed|failed|FAILED|Error|^[0-9.]+s " | tail -12 ) 2>&1 | tee $OUT/pytest_gpu.txt
