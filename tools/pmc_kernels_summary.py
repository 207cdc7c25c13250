"""Digest the PMC passes of tools/pmc_kernels.sh into pmc_traffic.json (the `traffic` field of bench.py's roofline entries).
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE tallies 64 B per 128-B request for wide
(16 B/lane) reads — all reads of these kernels — see MI355X_MICROARCH.md §HBM; WRITE_SIZE is taken as reported."""
import collections
import csv
import glob
import json
import os
import sys

out_dir, dtypes = sys.argv[1], sys.argv[2:]
csv.field_size_limit(1 << 30)
KEYS = {"mlp_fwd_kernel": None, "mlp_bwd_chain_kernel": "mlp_bwd_chain_kernel", "mlp_bwd_dw_f8_kernel": "mlp_bwd_dw_kernel",
        "mlp_bwd_dw_kernel": "mlp_bwd_dw_kernel", "mlp_bwd_reduce_kernel": "mlp_bwd_reduce_kernel"}
res = {}
for d in dtypes:
    for S in (192, 64, "merged"):
        P = 1024 * (192 + 64) if S == "merged" else 1024 * S
        vals = collections.defaultdict(dict)
        for C in ("FETCH_SIZE", "WRITE_SIZE"):
            fs = glob.glob(os.path.join(out_dir, "pmc_%s_%s_%s" % (d, S, C), "**", "*counter_collection.csv"), recursive=True)
            if not fs:
                continue
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(fs[0])):
                name = r["Kernel_Name"].replace("void ", "").replace("nerfhip::", "")
                per[(name.split("(")[0], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            agg = collections.defaultdict(list)
            for (name, _), cs in per.items():
                agg[name].append(cs.get(C, 0.0))
            for name, v in agg.items():
                vals[name][C] = sum(v) / len(v)
        for name, cs in vals.items():
            base = name.split("<")[0]
            if base not in KEYS or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
                continue
            if S == "merged" and base not in ("mlp_bwd_dw_f8_kernel", "mlp_bwd_dw_kernel", "mlp_bwd_reduce_kernel"):
                continue                       # (its forwards and chains are per-model launches, measured above)
            if base == "mlp_fwd_kernel":
                # template args <PREC, MODE, SIGMA_ONLY, SV>: SV 0 = inference, 1/2 = activation-saving
                args = name[name.index("<") + 1:name.rindex(">")].replace(" ", "").split(",")
                if args[2] == "true":
                    continue
                key = "mlp_fwd_kernel" if args[3] in ("0", "false") else "mlp_fwd_kernel<save>"
            else:
                key = KEYS[base]
            if S == "merged":
                key += "<merged>"
            res["%s|%s|%d" % (key, d, P)] = {
                "FETCH_SIZE_KB": round(cs["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(cs["WRITE_SIZE"], 1),
                "hbm_bytes_per_launch": int((2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024), "kernel": name,
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/kbench.py --dtype %s --samples %s" % (d, "192 --merged" if S == "merged" else S)}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_pl_amd.build import source_digest  # noqa: E402
out = dict(res)
out["_meta"] = {"source_digest": source_digest(), "note": "digest of nerf_pl_amd/csrc + include/nerfhip.h + build flags the counters were taken on"}
json.dump(out, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
for k, v in sorted(res.items()):
    print(k.ljust(44), "fetch %9.1f KB x2  write %9.1f KB  => %7.1f MB" % (v["FETCH_SIZE_KB"], v["WRITE_SIZE_KB"], v["hbm_bytes_per_launch"] / 1e6))
