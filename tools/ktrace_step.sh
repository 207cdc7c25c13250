#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the timed training step (bench.py) -> per (kernel, grid) durations.
#   tools/ktrace_step.sh <tag> [bench args...]
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python $REPO/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $OUT/bench_under_trace.json 2> $OUT/trace.log
cd $REPO
python - <<PY
import csv, collections
rows=list(csv.DictReader(open('$OUT/trace/t_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].replace('void ','').replace('nerfhip::','').split('(')[0][:44]
    d[(n,r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
with open('$OUT/kernel_by_grid.csv','w') as f:
    f.write('kernel,grid,calls,avg_us,min_us,median_us,max_us\n')
    for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
        v2=sorted(v)
        f.write('"%s",%s,%d,%.1f,%.1f,%.1f,%.1f\n'%(k[0],k[1],len(v),sum(v)/len(v),v2[0],v2[len(v2)//2],v2[-1]))
        if 'mlp' in k[0] or 'adam' in k[0] or 'fine_z' in k[0]:
            print(k[0].ljust(46), str(k[1]).rjust(8), len(v), 'avg %.1f min %.1f med %.1f'%(sum(v)/len(v), v2[0], v2[len(v2)//2]))
PY
find $OUT/trace -name "*.csv" -size +2M -delete
