# r06 call 33: last sanity of the in-tree library as it stands at the round's end: smoke, a parity / step subset, the driver bench command
set -u
OUT=gpurun_out/r06_33; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -n 2 $OUT/smoke.txt
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_step.py tests/test_gpu_training.py tests/test_gpu_render_fused.py tests/test_gpu_inference.py tests/test_bench_contract.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | cut -c1-300 ) 2>&1 | tee $OUT/pytest_subset.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_33/bench_default.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['value'], 'literal', d['literal_contract']['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'cpu', d['cpu_baseline']['value'])
PY
