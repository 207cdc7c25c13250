# r06 call 7: whole -m gpu suite at HEAD (18 minted reference seeds), the driver's bench command plain and under rocprofv3 --kernel-trace --stats
set -u
OUT=gpurun_out/r06_07; mkdir -p $OUT
( time timeout 3000 python -m pytest tests -q -m gpu --durations=8 -s 2>&1 | grep -E "passed|failed|FAILED|Error|^[0-9.]+s |render_rays 1024|coarse weights|PSNR vs reference|PSNR gate" | cut -c1-2500 ) 2>&1 | tee $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o drv -f csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_under_trace.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err
cd $GRAFT_REPO_ROOT
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/driver_cmd_kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -size +30M -delete
head -8 $OUT/driver_cmd_kernel_stats.csv
