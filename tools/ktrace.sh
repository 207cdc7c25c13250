#!/bin/bash
# per-kernel durations of tools/kbench.py under rocprofv3 for a given library: tools/ktrace.sh <tag> [lib.so]
TAG=$1; LIB=${2:-}
REPO=$(pwd); OUT=$REPO/gpurun_out/ktrace_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$LIB" ]; then export NERFHIP_LIB_PATH=$LIB; fi
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o t -- python $REPO/tools/kbench.py --reps 10 > $OUT/kbench.txt 2> $OUT/log.txt
cd $REPO
python - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].replace('void ','').replace('nerfhip::','')
    if 'mlp' in n: print('$TAG', n[:48], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1))
PY
