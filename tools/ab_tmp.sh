set -u
TAG=r02zz
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
timeout 200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 200 tools/ktrace_step.sh $TAG/trace_f8 > $OUT/kernel_by_grid_f8.txt
timeout 300 tools/pmc_kernels.sh $TAG/pmc bf16_f8 > $OUT/pmc_traffic.txt
timeout 100 python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_train_bf16_storage.json 2>/dev/null
tail -1 $OUT/bench_default.json | cut -c1-300
