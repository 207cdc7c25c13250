"""Digest rocprofv3 CSV output (kernel stats + PMC passes) into small per-kernel summaries.
usage: python tools/pmc_summary.py gpurun_out/prof_<tag> <tag>
Writes <dir>/<tag>_kernel_stats.csv, <dir>/<tag>_pmc_per_kernel.csv and <dir>/pmc_traffic.json.
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KB and, on
gfx950, FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) reads — every read of these
kernels (MI355X_MICROARCH.md §HBM)."""
import collections
import csv
import glob
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)


def find(sub, pat):
    fs = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def short(name):
    n = name.replace("void ", "").replace("nerfhip::", "")
    return n.split("(")[0][:90]


# ---- kernel stats (time) ----
ks = find("trace", "*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(d, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r.get("Name", "")), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"),
                        r.get("Percentage"), r.get("MinNs"), r.get("MaxNs")])
# per (kernel, grid) durations from the trace (separates the 64-sample coarse and 192-sample fine launches)
kt = find("trace", "*kernel_trace.csv")
dur = collections.defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt)):
        g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        dur[(short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(d, f"{tag}_kernel_by_grid.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Grid", "Calls", "AvgUs", "MinUs", "MaxUs"])
        for (k, g), v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            if "nerfhip" in k or "mlp" in k or "composite" in k or "fine_z" in k or "sample" in k:
                w.writerow([k, g, len(v), round(sum(v) / len(v) / 1e3, 2), round(min(v) / 1e3, 2), round(max(v) / 1e3, 2)])

# ---- PMC ----
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_l2"):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        continue
    per_disp = collections.defaultdict(float)
    meta = {}
    for r in csv.DictReader(open(cc)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_disp[key] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))
    for (disp, cname), val in per_disp.items():
        acc[meta[disp]][cname].append(val)
names = sorted({c for v in acc.values() for c in v})
with open(os.path.join(d, f"{tag}_pmc_per_kernel.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Grid", "Dispatches"] + names)
    for (k, g), cs in sorted(acc.items()):
        if not any(s in k for s in ("mlp", "composite", "fine_z", "sample", "posenc", "searchsorted")):
            continue
        n = max(len(v) for v in cs.values())
        w.writerow([k, g, n] + [round(sum(cs[c]) / len(cs[c]), 1) if c in cs else "" for c in names])

traffic = {}
for (k, g), cs in acc.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs and k.startswith("mlp_"):
        f_kb = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
        w_kb = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        traffic[f"{k}|grid={g}"] = {"FETCH_SIZE_KB": round(f_kb, 1), "WRITE_SIZE_KB": round(w_kb, 1),
                                    "hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024)}
json.dump(traffic, open(os.path.join(d, "pmc_traffic_raw.json"), "w"), indent=1)
print("pmc_summary: %d kernels with counters, %d with traffic" % (len(acc), len(traffic)))
