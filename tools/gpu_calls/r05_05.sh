# r05 call 5: the single-launch render kernels, first time on hardware (bit-equality tests under a timeout), then the gradient
# tests whose bounds changed, then the default bench line
set -u
OUT=gpurun_out/r05_05; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_render_fused.py -q -m gpu -x > $OUT/pytest_render.txt 2>&1; echo "render tests rc $?"
grep -E "passed|failed|^FAILED|^E  " $OUT/pytest_render.txt | tail -12
timeout 600 python -m pytest tests/test_gpu_training.py "tests/test_gpu_bf16.py::test_timed_node_at_benchmark_size_vs_oracle_gradients" tests/test_gpu_parity.py tests/test_gpu_fused_step.py tests/test_gpu_draws.py -q -m gpu -s > $OUT/pytest_other.txt 2>&1
grep -E "passed|failed|^FAILED|^E  |fine pass on identical" $OUT/pytest_other.txt | tail -12
( time timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r05_05/bench.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], {k: d.get(k) for k in ('launches_per_step','non_mlp_us','mlp_kernels_us_per_step','step_frac_mfma','render_fwd_rays_per_s_per_gpu','eval_ms_per_image')})
    print(d['roofline'])
except Exception as e: print('bench failed', e); print(open('gpurun_out/r05_05/bench.err').read()[-1500:])
PY
