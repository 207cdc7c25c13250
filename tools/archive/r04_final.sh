# Round-4 evidence set (run ON THE GPU BOX through gpurun): what profiles/README.md "Round 4" quotes, one box, one call
#   tools/r04_final.sh [tag]      SKIP_TESTS=1 skips the GPU test suite
set -u
TAG=${1:-r04final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
REPO=$(pwd)
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tools/ktrace_step.sh $TAG/trace > $OUT/kernel_by_grid.txt 2>&1
python bench.py --dtype bf16_f8 --no-cpu-baseline --no-extras > $OUT/bench_train_f8_dw.json 2>/dev/null
python bench.py --dtype fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_train_fp32.json 2>/dev/null
python bench.py --mode render --no-cpu-baseline > $OUT/bench_render.json 2>/dev/null
python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_eval.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --modular-step > $OUT/bench_modular_step.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --fuse-adam > $OUT/bench_fuse_adam.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --force-dist --sync-in-graph 1 > $OUT/bench_rccl_world1_one_graph.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --force-dist --sync-in-graph 0 > $OUT/bench_rccl_world1_two_graphs.json 2>/dev/null
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('launches_per_step', 'non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')})
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
[ -n "${SKIP_PROBES:-}" ] || { timeout 100 python tools/small_kernel_bench.py > $OUT/small_kernels.txt 2>&1; tail -2 $OUT/small_kernels.txt; }
[ -n "${SKIP_PROBES:-}" ] || timeout 60 tools/probes/probe_mall.bin > $OUT/probe_mall.txt 2>&1
[ -n "${SKIP_TESTS:-}" ] || ( time python -m pytest tests -q -m gpu --durations=8 2>&1 | grep -E "passed|failed|FAILED|Error|^[0-9.]+s " | tail -14 ) 2>&1 | tee $OUT/pytest_gpu.txt
ls $OUT
