mkdir -p gpurun_out/r03c
python -m pytest tests/test_gpu_fused_step.py -x -q -s 2>&1 | tail -40 > gpurun_out/r03c/pytest_fused.log
tail -25 gpurun_out/r03c/pytest_fused.log
python -m pytest tests -q -m gpu --deselect tests/test_gpu_psnr_gate.py --deselect tests/test_gpu_fused_step.py 2>&1 | tail -15 > gpurun_out/r03c/pytest_rest.log
tail -15 gpurun_out/r03c/pytest_rest.log
for v in "" "--no-fuse-adam" "--modular-step" "--force-dist --sync-in-graph 0" "--force-dist --sync-in-graph 1"; do
  tag=$(echo "default$v" | tr -d ' -')
  python bench.py --no-extras --no-cpu-baseline $v > gpurun_out/r03c/bench_$tag.json 2> gpurun_out/r03c/bench_$tag.err
  python - "$tag" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r03c/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['ms_per_step'], d['value'], d.get('mlp_kernels_us_per_step'), d['config'].get('step_form'), d['config'].get('grad_sync'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open('gpurun_out/r03c/bench_%s.err'%sys.argv[1]).read()[-1500:])
PY
done
