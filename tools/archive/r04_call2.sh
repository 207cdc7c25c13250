#!/bin/bash
OUT=gpurun_out/r04c2; mkdir -p $OUT
python tools/probes/probe_randn.py > $OUT/probe_randn.txt 2>&1; cat $OUT/probe_randn.txt | tail -40
timeout 600 python -m pytest tests/test_gpu_draws.py -q > $OUT/pytest_draws.txt 2>&1; echo "draws rc=$?"; grep -E "passed|failed|FAILED" $OUT/pytest_draws.txt | tail -30
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $REPO/$OUT/trace -o t -- python $REPO/bench.py --no-extras --no-cpu-baseline --steps 40 > $REPO/$OUT/bench_traced.json 2> $REPO/$OUT/bench_traced.err
cd $REPO
python tools/rocpd_stats.py $OUT/trace 2>/dev/null | head -5
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); echo $f; head -40 $f | cut -c1-180
find $OUT/trace -name "*.csv" -size +2M -delete
