# r06 call 25: whole -m gpu suite at the regen commit (default off) with the 50-run reference fixture (36 live seeds), driver bench command plain + under rocprofv3, smoke, PMC traffic
# (by-grid digest), smoke(), PMC FETCH_SIZE / WRITE_SIZE passes of the MLP kernels
set -u
OUT=gpurun_out/r06_25; mkdir -p $OUT
( time timeout 3000 python -m pytest tests -q -m gpu --durations=8 -s 2>&1 | grep -E "passed|failed|FAILED|Error|^[0-9.]+s |render_rays 1024|coarse weights|PSNR vs reference|PSNR gate" | cut -c1-2500 ) 2>&1 | tee $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
tools/ktrace_step.sh r06_25/trace > $OUT/kernel_by_grid.txt; cat $OUT/kernel_by_grid.txt
cp $OUT/trace/trace/t_kernel_stats.csv $OUT/driver_cmd_kernel_stats.csv 2>/dev/null; head -12 $OUT/driver_cmd_kernel_stats.csv
tools/pmc_kernels.sh r06_25/pmc bf16 > $OUT/pmc_traffic.txt 2>&1; tail -20 $OUT/pmc_traffic.txt
