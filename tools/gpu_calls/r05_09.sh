# r05 call 9 (re-entry: the evidence files of calls 5-8 were lost with the container): the round's evidence set on one box
#   default line (+ time), the same command under rocprofv3 --kernel-trace --stats, A/Bs (single-launch forward on/off,
#   Adam in the reduce), the other modes, eval trace, the whole -m gpu suite
set -u
OUT=gpurun_out/r05_09; mkdir -p $OUT
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tools/ktrace_step.sh r05_09/trace > $OUT/kernel_by_grid.txt 2>&1
{
for i in 1 2; do
  echo "single launch ON  $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 200 --warmup 20 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("launches_per_step"))')"
  echo "single launch OFF $(NERFHIP_RENDER_FUSED=0 python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 200 --warmup 20 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("launches_per_step"))')"
done
} > $OUT/ab_single_launch_forward.txt 2>&1
{
for i in 1 2; do
  echo "separate Adam   $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 200 --warmup 20 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("launches_per_step"))')"
  echo "Adam in reduce  $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 200 --warmup 20 --fuse-adam | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("launches_per_step"))')"
done
} > $OUT/ab_adam_in_reduce.txt 2>&1
python bench.py --dtype bf16_f8 --no-cpu-baseline --no-extras > $OUT/bench_train_f8_dw.json 2>/dev/null
python bench.py --dtype fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_train_fp32.json 2>/dev/null
python bench.py --mode render --no-cpu-baseline > $OUT/bench_render.json 2>/dev/null
python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_eval.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$OUT/eval_trace -o t -- python $GRAFT_REPO_ROOT/bench.py --mode eval --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 )
python - <<'PY' > $OUT/eval_trace_by_kernel.txt 2>&1
import csv, collections, glob
f = glob.glob('gpurun_out/r05_09/eval_trace/**/t_kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('void ', '').replace('nerfhip::', '').split('(')[0][:60]
    d[(n, r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('kernel,grid,calls,avg_us,min_us,median_us,max_us')
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:24]:
    v2 = sorted(v); print('"%s",%s,%d,%.1f,%.1f,%.1f,%.1f' % (k[0], k[1], len(v), sum(v) / len(v), v2[0], v2[len(v2) // 2], v2[-1]))
PY
find $OUT/eval_trace -name "*.csv" -size +2M -delete
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('launches_per_step', 'non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')})
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
cat $OUT/ab_single_launch_forward.txt $OUT/ab_adam_in_reduce.txt; tail -12 $OUT/kernel_by_grid.txt; head -8 $OUT/eval_trace_by_kernel.txt
( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | grep -E "passed|failed|FAILED|Error|^[0-9.]+s " | tail -14 ) 2>&1 | tee $OUT/pytest_gpu.txt
