#!/bin/bash
# Run ON THE GPU BOX: where do the waves of the activation-saving forward (and the chain) wait?  One rocprofv3 --pmc pass
# (--kernel-trace only) per counter group over tools/kbench.py; digest -> gpurun_out/<tag>/pmc_stall.txt
#   tools/pmc_stall.sh <tag> [dtype]
TAG=$1; D=${2:-bf16_f8}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
CGRP=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"
 "SQ_IFETCH SQ_IFETCH_LEVEL"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_STALL"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WRITE_WAVEFRONTS_sum"
 "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_sum"
 "MemUnitStalled GRBM_UTCL2_BUSY TCC_TAG_STALL_sum TCC_IB_STALL_sum"
 "TCP_TCP_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum"
 "TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum"
 "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum"
 "TCC_REQ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"
)
i=0
for G in "${CGRP[@]}"; do
  timeout 90 rocprofv3 --pmc $G --kernel-trace -f csv -d $OUT/g$i -o p -- python $REPO/tools/kbench.py --dtype $D --reps 2 > /dev/null 2> $OUT/g$i.log || echo "group $i failed: $G" >> $OUT/failed.txt
  i=$((i+1))
done
cd $REPO
python - <<PY > $OUT/pmc_stall.txt
import csv, glob, collections
csv.field_size_limit(1 << 30)
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("nerfhip::", "").split("(")[0]
        if "mlp_" not in n or r["Grid_Size"] not in ("393216",):
            continue
        per[(n, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (n, _), cs in per.items():
        for c, v in cs.items():
            tot[n][c].append(v)
for n in sorted(tot):
    print(n)
    for c in sorted(tot[n]):
        v = tot[n][c]
        print("   %-44s %16.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
cat $OUT/pmc_stall.txt | head -120
find $OUT -name "*.csv" -size +1M -delete
