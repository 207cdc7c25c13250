# r06 call 30: the whole -m gpu suite, smoke and the driver bench command once more at HEAD (the in-tree library as rebuilt after the reverted experiments of calls 26-29)
# (by-grid digest), smoke(), PMC FETCH_SIZE / WRITE_SIZE passes of the MLP kernels
set -u
OUT=gpurun_out/r06_30; mkdir -p $OUT
( time timeout 3000 python -m pytest tests -q -m gpu --durations=8 -s 2>&1 | grep -E "passed|failed|FAILED|Error|^[0-9.]+s |render_rays 1024|coarse weights|PSNR vs reference|PSNR gate" | cut -c1-2500 ) 2>&1 | tee $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
tools/ktrace_step.sh r06_30/trace > $OUT/kernel_by_grid.txt; cat $OUT/kernel_by_grid.txt
cp $OUT/trace/trace/t_kernel_stats.csv $OUT/driver_cmd_kernel_stats.csv 2>/dev/null; head -12 $OUT/driver_cmd_kernel_stats.csv
