#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
#   tools/profile_gpu.sh r01 [bench args...]
# Writes gpurun_out/prof_<tag>/...; copy the summaries you want judged into profiles/ afterwards
# (tools/pmc_summary.py does that for the CSV digests).
# Counters are collected in their own runs, never combined with tracing other than --kernel-trace.
set -u
TAG=${1:-r00}; shift || true
ARGS="${@:---steps 20 --warmup 5 --no-cpu-baseline}"
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel time
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.log
# 2. PMC passes (few steps; counter collection serialises dispatches)
PARGS="--steps 3 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o pmc -- python $REPO/bench.py $PARGS > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o pmc -- python $REPO/bench.py $PARGS > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq -o pmc -- python $REPO/bench.py $PARGS > /dev/null 2> $OUT/pmc_sq.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/pmc_l2 -o pmc -- python $REPO/bench.py $PARGS > /dev/null 2> $OUT/pmc_l2.log
cd $REPO
python tools/pmc_summary.py $OUT $TAG
find $OUT -name '*.csv' -size +2M -delete   # keep the pull small; digests are already written
ls -la $OUT
