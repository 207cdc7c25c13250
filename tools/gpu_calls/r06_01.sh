# r06 call 1: the dW bisect staircase (tools/probes/dw_bisect.hip, skeleton -> product shape, zeros and random data) + this box's baseline line
set -u
OUT=gpurun_out/r06_01; mkdir -p $OUT
rocm-smi --showpower --showclocks > $OUT/smi_idle.txt 2>&1
timeout 300 tools/probes/bin/dw_bisect 20 > $OUT/dw_bisect.txt 2>&1
tail -70 $OUT/dw_bisect.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err ) 2> $OUT/bench_driver_cmd.time
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_01/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('cold_start_ms_per_step'), d.get('roofline'))
print(d.get('roofline_kernels'))
PY
