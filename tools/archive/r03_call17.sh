mkdir -p gpurun_out/r03q
OUT=gpurun_out/r03q
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --dtype bf16 --no-cpu-baseline --no-extras > $OUT/bench_train_bf16_storage.json 2>/dev/null
python bench.py --dtype fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_train_fp32.json 2>/dev/null
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['dtype'], d.get('roofline_north_star',{}).get('avg_launch_us'))
PY
done
