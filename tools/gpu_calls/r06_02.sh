# r06 call 2: dW bisect v2 (epilogue / bias variants); new tests (single-launch test_time, bench contract); r4 / r5 / r6 builds under ONE
# harness (literal contract vs sustained); PSNR vs the minted reference curves (seeds available so far, both row-total roundings);
# eval with / without the single-launch test_time; the N>1 step forms at world 1 over RCCL; by-grid traces of configs[2] and configs[3]
set -u
OUT=gpurun_out/r06_02; mkdir -p $OUT
timeout 300 tools/probes/bin/dw_bisect 20 > $OUT/dw_bisect.txt 2>&1; grep "rep 1 data random" $OUT/dw_bisect.txt
( timeout 900 python -m pytest tests/test_gpu_render_fused.py tests/test_bench_contract.py tests/test_gpu_inference.py -q -m gpu -x 2>&1 | tail -8 ) | tee $OUT/pytest_new.txt
# ---- same-harness comparison of the three builds: literal contract (W=5 untimed + K=20 timed right after the build) and sustained
for T in r04 r05; do
  ( cd .oldtrees/$T && python bench.py --gpus 1 --steps 20 --warmup 9 --no-cpu-baseline --no-extras > ../../$OUT/harness_${T}_literal.json 2>/dev/null )
  ( cd .oldtrees/$T && python bench.py --gpus 1 --steps 20 --warmup 159 --no-cpu-baseline --no-extras > ../../$OUT/harness_${T}_sustained.json 2>/dev/null )
done
( cd .oldtrees/r05 && python bench.py --gpus 1 --steps 20 --warmup 5 --settle 0 --no-cpu-baseline --no-extras --no-pmc > ../../$OUT/harness_r05_literal_settle0.json 2>/dev/null )
( cd .oldtrees/r05 && python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > ../../$OUT/harness_r05_default.json 2>/dev/null )
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > $OUT/harness_r06_default.json 2>/dev/null
for f in $OUT/harness_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'ms/step', d['ms_per_step'], 'warmup', d['warmup'], 'cold', d.get('cold_start_ms_per_step'), 'literal', (d.get('literal_contract') or {}).get('ms_per_step'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
# ---- PSNR vs the reference's own training runs
timeout 900 python tests/tools/psnr_vs_reference.py --out $OUT/psnr_vs_reference.json 2>&1 | grep -v "^seed" | cut -c1-1500 | tee $OUT/psnr_vs_reference.txt
# ---- eval: single-launch test_time on / off
NERFHIP_FUSE_TEST_TIME=1 python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_eval_fused.json 2>/dev/null
NERFHIP_FUSE_TEST_TIME=0 python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_eval_launches.json 2>/dev/null
NERFHIP_FUSE_TEST_TIME=1 python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_eval_fused2.json 2>/dev/null
NERFHIP_FUSE_TEST_TIME=0 python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_eval_launches2.json 2>/dev/null
# ---- the N>1 step forms at world 1 over RCCL
python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/dist_plain.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --no-pmc --force-dist > $OUT/dist_merged_one_graph.json 2>$OUT/dist_merged_one_graph.err
python bench.py --no-cpu-baseline --no-extras --no-pmc --force-dist --grad-sync-form per_model > $OUT/dist_per_model_one_graph.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --no-pmc --force-dist --sync-in-graph 0 > $OUT/dist_merged_two_graphs.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/dist_plain2.json 2>/dev/null
for f in $OUT/bench_eval*.json $OUT/dist_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], 'ms/step', d['ms_per_step'], 'launches', d.get('launches_per_step'), d['config'].get('grad_sync'), d['config'].get('capture_fallback'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
tail -3 $OUT/dist_merged_one_graph.err
# ---- traces
tools/ktrace_step.sh r06_02/trace_c2 --no-extras --no-pmc > $OUT/kernel_by_grid_c2.txt 2>&1; tail -12 $OUT/kernel_by_grid_c2.txt
tools/ktrace_step.sh r06_02/trace_c3 --no-extras --no-pmc --workload c3 > $OUT/kernel_by_grid_c3.txt 2>&1; tail -12 $OUT/kernel_by_grid_c3.txt
python - <<'PY'
import json
for t in ('c2','c3'):
    try:
        d=json.loads(open('gpurun_out/r06_02/trace_%s/bench_under_trace.json'%t).read().strip().splitlines()[-1])
        print(t, d['ms_per_step'], d.get('step_frac_mfma'), [(k['kernel'][:24], k['in_step_launch_us'], k['avg_launch_us']) for k in d.get('roofline_kernels', [])])
    except Exception as e: print(t, 'FAILED', e)
PY
