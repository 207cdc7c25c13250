#!/bin/bash
OUT=gpurun_out/r04c9; mkdir -p $OUT
REPO=$(pwd)
for D in bf16_f8 bf16; do for S in 192 64; do
  python tools/kbench.py --dtype $D --samples $S --reps 20 2>/dev/null | tail -1 | tee -a $OUT/kbench_small_bound.txt
  NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_small.so python tools/kbench.py --dtype $D --samples $S --reps 20 2>/dev/null | tail -1 | tee -a $OUT/kbench_small_bound.txt
done; done
( time timeout 1200 python -m pytest tests -q -m gpu --durations=6 > $OUT/pytest_gpu.txt 2>&1 ) 2> $OUT/pytest_gpu.time; echo "gpu tests rc=$?"; grep -E "passed|failed|FAILED|^[0-9.]+s |HIP vs oracle" $OUT/pytest_gpu.txt | tail -14; grep real $OUT/pytest_gpu.time
