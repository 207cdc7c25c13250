#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the round's measured evidence in one go.
#   tools/profile_gpu.sh <tag>       -> gpurun_out/<tag>/...   (copy what should be judged into profiles/)
#   1. python bench.py (default: the driver's command, with cpu_baseline)        -> bench_default.json
#   2. the same under rocprofv3 --kernel-trace --stats                            -> trace/, kernel_by_grid.csv
#   3. bench --dtype bf16 (bf16 storage), --mode render, --mode eval, --dtype fp32 -> bench_*.json
#   4. PMC FETCH_SIZE / WRITE_SIZE passes of the MLP kernels (tools/pmc_kernels.sh) -> pmc_traffic.json
# Counters are collected in their own runs, never combined with tracing other than --kernel-trace.
set -u
TAG=${1:-r00}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tools/ktrace_step.sh $TAG/trace_f8 > $OUT/kernel_by_grid_f8.txt
python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_train_bf16_storage.json 2>/dev/null
python bench.py --mode render --no-cpu-baseline > $OUT/bench_render.json 2>/dev/null
python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_eval.json 2>/dev/null
python bench.py --dtype fp32 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_train_fp32.json 2>/dev/null
tools/pmc_kernels.sh $TAG/pmc bf16_f8 bf16 > $OUT/pmc_traffic.txt
python tools/kbench_hbm.py > $OUT/hbm_kernels.txt 2>/dev/null
ls $OUT
