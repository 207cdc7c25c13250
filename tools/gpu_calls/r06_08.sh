# r06 call 8: SQ issue accounts of the round-6 kernels (one --pmc pass per counter group, --kernel-trace only), the driver's literal command, smoke()
set -u
OUT=gpurun_out/r06_08; mkdir -p $OUT
tools/pmc_issue.sh r06_08/pmc bf16 > $OUT/pmc_issue.log 2>&1
cp $OUT/pmc/pmc_issue.txt $OUT/pmc_issue.txt 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
tail -c 400 $OUT/bench_driver_cmd.json; tail -3 $OUT/smoke.txt; head -12 $OUT/pmc_issue.txt | cut -c1-300
