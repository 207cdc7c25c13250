#!/bin/bash
# Run ON THE GPU BOX: what do the waves of the step's MLP kernels spend their cycles on?  One rocprofv3 --pmc pass (--kernel-trace
# only) per counter group over `bench.py --pmc-launch` (every MLP kernel of the step twice on resident buffers, incl. the
# single-launch forward and the inference forward); digest -> gpurun_out/<tag>/pmc_issue.txt
#   tools/pmc_issue.sh <tag> [dtype]
# Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES
# counts cycles (32 per v_mfma_f32_32x32x16_bf16) summed over SIMDs; GRBM_GUI_ACTIVE = shader-clock cycles of the launch.
TAG=$1; D=${2:-bf16}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CGRP=(
 "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS"
 "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
)
i=0
for G in "${CGRP[@]}"; do
  timeout 120 rocprofv3 --pmc $G --kernel-trace -f csv -d $OUT/g$i -o p -- python $REPO/bench.py --pmc-launch --dtype $D > /dev/null 2> $OUT/g$i.log || echo "group $i failed: $G" >> $OUT/failed.txt
  i=$((i+1))
done
cd $REPO
python tools/pmc_issue_summary.py $OUT $D > $OUT/pmc_issue.txt
cat $OUT/failed.txt 2>/dev/null
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
