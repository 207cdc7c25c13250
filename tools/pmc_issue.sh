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
python - <<PY > $OUT/pmc_issue.txt
import csv, glob, collections
csv.field_size_limit(1 << 30)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("nerfhip::", "").split("(")[0]
        if "mlp_" not in n or "pack" in n:
            continue
        per[(n, r["Grid_Size"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (n, g, _), cs in per.items():
        for c, v in cs.items():
            vals[(n, g)][c].append(v)
print("counters per launch (mean over the launches of one `bench.py --pmc-launch --dtype $D`); fractions of the waves' own cycles unless said otherwise")
for (n, g), cs in sorted(vals.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    if not wc or not gui:
        continue
    fr = lambda c: (m.get(c, float("nan")) / wc)
    print("%s grid %s" % (n, g))
    print("   GRBM_GUI_ACTIVE %.3e cycles; MFMA pipe busy %.3f of 1024 SIMDs x GUI cycles (SQ_VALU_MFMA_BUSY_CYCLES %.3e, SQ_INSTS_MFMA %.3e)"
          % (gui, m.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (1024 * gui), m.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")), m.get("SQ_INSTS_MFMA", float("nan"))))
    print("   of SQ_WAVE_CYCLES (%.3e quad-cycles, %d waves): issuing any %.3f [VALU+MFMA %.3f, LDS %.3f, VMEM %.3f, FLAT %.3f, SALU %.3f, misc %.3f]; waiting on an instruction %.3f, waiting at all %.3f"
          % (wc, m.get("SQ_WAVES", 0), fr("SQ_ACTIVE_INST_ANY"), fr("SQ_ACTIVE_INST_VALU"), fr("SQ_ACTIVE_INST_LDS"), fr("SQ_ACTIVE_INST_VMEM"), fr("SQ_ACTIVE_INST_FLAT"),
             fr("SQ_ACTIVE_INST_SCA"), fr("SQ_ACTIVE_INST_MISC"), fr("SQ_WAIT_INST_ANY"), fr("SQ_WAIT_ANY")))
    nw = max(m.get("SQ_WAVES", 1), 1)
    args = tuple(m.get(c, float("nan")) / nw for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD"))
    args += (m.get("SQ_LDS_BANK_CONFLICT", float("nan")) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1),)
    print("   instructions per wave: VALU %.0f (of which MFMA %.0f), SALU %.0f, LDS %.0f, VMEM write %.0f / read %.0f; LDS bank-conflict cycles / LDS active %.4f" % args)
PY
cat $OUT/failed.txt 2>/dev/null
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
