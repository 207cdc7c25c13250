# r05 call 10: per-replay durations of the timed steps at the driver's settings (why is --steps 20 --warmup 5 slower than 200 steps?)
set -u
OUT=gpurun_out/r05_10; mkdir -p $OUT
python bench.py --steps 150 --warmup 5 --series --no-extras --no-cpu-baseline --no-pmc > $OUT/series_w5.json 2> $OUT/series_w5.err
python bench.py --steps 150 --warmup 5 --series --no-extras --no-cpu-baseline --no-pmc > $OUT/series_w5_b.json 2> $OUT/series_w5_b.err
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/plain_20_5.json 2>/dev/null
python bench.py --steps 20 --warmup 100 --no-extras --no-cpu-baseline --no-pmc > $OUT/plain_20_100.json 2>/dev/null
grep "per-step" $OUT/series_w5.err | cut -c1-1500
python - <<'PY'
import json
for f in ('series_w5','series_w5_b','plain_20_5','plain_20_100'):
    d=json.loads(open('gpurun_out/r05_10/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'])
PY
