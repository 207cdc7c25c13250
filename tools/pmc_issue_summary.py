"""Digest of tools/pmc_issue.sh's counter passes (rocprofv3 --pmc, one group per pass): per kernel, where the waves' cycles go.
    python tools/pmc_issue_summary.py <dir with g0..gN> [dtype]
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES
counts cycles (32 per v_mfma_f32_32x32x16_bf16) summed over SIMDs; GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (one GRBM each), so
the launch's shader cycles = GUI / 8 — including ~10 us of dispatch overhead per launch (the 15 us reduce kernel "runs at 3.4 GHz" by
this count): the clock it implies is an upper bound, ~4 % high for a 300 us kernel, and the MFMA-busy fraction a lower bound."""
import collections
import csv
import glob
import sys

out, dtype = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bf16")
csv.field_size_limit(1 << 30)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("nerfhip::", "").split("(")[0]
        if "mlp_" not in n or "pack" in n:
            continue
        per[(n, r["Grid_Size"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        if "/g0/" in f and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            durs[(n, r["Grid_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (n, g, _), cs in per.items():
        for c, v in cs.items():
            vals[(n, g)][c].append(v)
print("counters per launch (mean over the launches of one `bench.py --pmc-launch --dtype %s`, each launch alone with the profiler's gaps around it);" % dtype)
print("fractions of the waves' own cycles unless said otherwise")
for (n, g), cs in sorted(vals.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    if not wc or not gui:
        continue
    nan = float("nan")
    fr = lambda c: (m.get(c, nan) / wc)  # noqa: E731
    cyc = gui / 8.0
    us = sum(durs[(n, g)]) / len(durs[(n, g)]) if durs.get((n, g)) else nan
    print("%s grid %s" % (n, g))
    print("   %.1f us per launch under the counter pass; GRBM_GUI_ACTIVE / 8 = %.3e cycles (shader clock <= %.2f GHz); MFMA pipe busy >= %.3f of 1024 SIMDs x "
          "those cycles (SQ_VALU_MFMA_BUSY_CYCLES %.3e, SQ_INSTS_MFMA %.3e)"
          % (us, cyc, cyc / us / 1e3, m.get("SQ_VALU_MFMA_BUSY_CYCLES", nan) / (1024 * cyc), m.get("SQ_VALU_MFMA_BUSY_CYCLES", nan), m.get("SQ_INSTS_MFMA", nan)))
    print("   of SQ_WAVE_CYCLES (%.3e quad-cycles, %d waves): issuing any %.3f [VALU+MFMA %.3f, LDS %.3f, VMEM %.3f, FLAT %.3f, SALU %.3f, misc %.3f]; "
          "waiting on an instruction %.3f, waiting at all %.3f"
          % (wc, m.get("SQ_WAVES", 0), fr("SQ_ACTIVE_INST_ANY"), fr("SQ_ACTIVE_INST_VALU"), fr("SQ_ACTIVE_INST_LDS"), fr("SQ_ACTIVE_INST_VMEM"),
             fr("SQ_ACTIVE_INST_FLAT"), fr("SQ_ACTIVE_INST_SCA"), fr("SQ_ACTIVE_INST_MISC"), fr("SQ_WAIT_INST_ANY"), fr("SQ_WAIT_ANY")))
    nw = max(m.get("SQ_WAVES", 1), 1)
    args = tuple(m.get(c, nan) / nw for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD"))
    args += (m.get("SQ_LDS_BANK_CONFLICT", nan) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1),)
    print("   instructions per wave: VALU %.0f (of which MFMA %.0f), SALU %.0f, LDS %.0f, VMEM write %.0f / read %.0f; LDS bank-conflict cycles / LDS active %.4f" % args)
