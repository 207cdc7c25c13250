# r05 call 7: full GPU suite on the current build; ABBA A/B of Adam inside the reduce kernel; fine_z with the in-order fast path
set -u
OUT=gpurun_out/r05_07; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | grep -E "passed|failed|FAILED|Error|^[0-9.]+s " | tail -16 ) 2>&1 | tee $OUT/pytest_gpu.txt
b() { tag=$1; shift; "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2].ljust(28), d['value'], d['ms_per_step'], {k: d.get(k) for k in ('launches_per_step','non_mlp_us','mlp_kernels_us_per_step')})
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
A="--no-cpu-baseline --no-extras --no-pmc --steps 200 --warmup 20"
for i in 1 2; do
b sep_adam_a$i   python bench.py $A
b fuse_adam_a$i  python bench.py $A --fuse-adam
b fuse_adam_b$i  python bench.py $A --fuse-adam
b sep_adam_b$i   python bench.py $A
done
timeout 100 python tools/small_kernel_bench.py > $OUT/small_kernels.txt 2>&1; tail -3 $OUT/small_kernels.txt
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $REPO/$OUT/evaltrace -o t -- python $REPO/bench.py --mode eval --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $REPO; python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/r05_07/evaltrace/t_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].replace('void ','').replace('nerfhip::','').split('(')[0][:44]
    d[(n,r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:8]:
    print(k[0].ljust(46), str(k[1]).rjust(9), len(v), 'avg %.1f min %.1f'%(sum(v)/len(v), min(v)))
PY
find $OUT/evaltrace -name "*.csv" -size +1M -delete
