#!/bin/bash
# Run ON THE GPU BOX: HBM traffic (rocprofv3 PMC FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-trace only) of the MLP
# kernels of the training step, each launched alone by tools/kbench.py at the benchmark sizes.
#   tools/pmc_kernels.sh <tag> [dtype ...]        -> gpurun_out/<tag>/pmc_traffic.json (+ per-kernel CSV)
TAG=$1; shift
DTYPES=${@:-bf16_f8 bf16}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for D in $DTYPES; do
  for S in 192 64; do
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout 150 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/pmc_${D}_${S}_$C -o p -- python $REPO/tools/kbench.py --dtype $D --samples $S --reps 2 > /dev/null 2> $OUT/pmc_${D}_${S}_$C.log
    done
  done
  for C in FETCH_SIZE WRITE_SIZE; do          # the merged dW / reduce launches of the fused step (fine 192 + coarse 64 samples per ray)
    timeout 150 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/pmc_${D}_merged_$C -o p -- python $REPO/tools/kbench.py --dtype $D --samples 192 --merged --reps 2 > /dev/null 2> $OUT/pmc_${D}_merged_$C.log
  done
done
cd $REPO
python tools/pmc_kernels_summary.py $OUT $DTYPES
find $OUT -name "*.csv" -size +1M -delete
