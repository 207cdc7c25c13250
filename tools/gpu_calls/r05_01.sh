# r05 call 1: HEAD of round 4 re-checked on hardware (the prologue rework of 4bd5fd0 was never run on a GPU)
set -u
OUT=gpurun_out/r05_01; mkdir -p $OUT
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
timeout 600 python -m pytest tests/test_gpu_draws.py tests/test_gpu_fused_step.py -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest_subset.txt
timeout 100 python tools/small_kernel_bench.py > $OUT/small_kernels.txt 2>&1; tail -12 $OUT/small_kernels.txt
tools/ktrace_step.sh r05_01/trace > $OUT/kernel_by_grid.txt 2>&1; tail -25 $OUT/kernel_by_grid.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_01/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: d.get(k) for k in ('launches_per_step','non_mlp_us','mlp_kernels_us_per_step','step_frac_mfma')})
for k in d['roofline_kernels']: print(k['kernel'][:50], k['avg_launch_us'])
PY
