# r05 call 19: SQ counters of the small eval kernels (fine_z, compositing)
set -u
OUT=$PWD/gpurun_out/r05_19; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  timeout 120 rocprofv3 --pmc $G --kernel-trace -f csv -d $OUT/g$i -o p -- python $GRAFT_REPO_ROOT/bench.py --mode eval --steps 1 --warmup 0 --settle 0 --no-cpu-baseline > /dev/null 2> $OUT/g$i.log
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
csv.field_size_limit(1 << 30)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/r05_19/g*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("nerfhip::", "").split("(")[0]
        if not any(k in n for k in ("fine_z", "composite")): continue
        per[(n, r["Grid_Size"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (n, g, _), cs in per.items():
        for c, v in cs.items(): vals[(n, g)][c].append(v)
for (n, g), cs in sorted(vals.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    nw = max(m.get("SQ_WAVES", 1), 1); wc = max(m.get("SQ_WAVE_CYCLES", 1), 1)
    print(n, g, "waves %d" % nw, "per wave:", {c: round(m.get(c, 0) / nw, 1) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES")},
          "of wave cycles:", {c: round(m.get(c, 0) / wc, 3) for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
          "GUI/8 cycles %.3e" % (m.get("GRBM_GUI_ACTIVE", 0) / 8), "bank conflict share %.3f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
find gpurun_out/r05_19 -name "*.csv" -size +1M -delete
