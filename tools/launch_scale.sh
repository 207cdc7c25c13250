#!/bin/bash
# Multi-GPU scaling run of bench.py on ONE node (the driver's SCALE protocol): N = 1, 2, 4, 8 ranks, one per GPU, over
# RCCL/xGMI — ONE command for the first box that has more than one GPU:
#     tools/launch_scale.sh [outdir] [extra bench args...]        -> <outdir>/SCALE.json (+ n<N>.json / n<N>.err per run)
# `python bench.py --gpus N` launches its own N ranks (torch.distributed.run) and refuses to print a line for any other world
# size; this script additionally checks each line (n_gpus == N, config.rccl_nranks == N), RCCL's own INIT log, and writes the
# per-N lines with the weak-scaling efficiency value(N) / (N * value(1)) into one JSON.  The N > 1 step defaults to two graphs
# with eager all-reduces; NERFHIP_SYNC_IN_GRAPH=1 tools/launch_scale.sh ... measures the one-graph form (run both: the one-graph
# capture with RCCL collectives inside has never met a real multi-rank communicator).
OUT=${1:-gpurun_out/scale}; shift || true
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=${NCCL_DEBUG:-INFO} NCCL_DEBUG_SUBSYS=INIT
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
RAN=""
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "[scale] only $NGPU GPU(s) visible: skipping N=$N"; continue; fi
  python bench.py --gpus $N --steps 50 --warmup 10 --no-cpu-baseline "$@" > $OUT/n$N.json 2> $OUT/n$N.err || { echo "[scale] N=$N failed (rc $?)"; tail -5 $OUT/n$N.err; continue; }
  RAN="$RAN $N"
done
python - "$OUT" "$NGPU" $RAN <<'PY'
import json, os, re, sys
out, ngpu, ns = sys.argv[1], int(sys.argv[2]), [int(x) for x in sys.argv[3:]]
runs, problems = [], []
for n in ns:
    line = [l for l in open(os.path.join(out, "n%d.json" % n)) if l.startswith("{")]
    if not line:
        problems.append("no JSON line for N=%d" % n)
        continue
    d = json.loads(line[-1])
    if d["n_gpus"] != n:
        problems.append("N=%d: line says n_gpus %s" % (n, d["n_gpus"]))
    if n > 1:
        if d["config"]["rccl_nranks"] != n:
            problems.append("N=%d: rccl_nranks %s" % (n, d["config"]["rccl_nranks"]))
        ranks = set(re.findall(r"nranks (\d+)", open(os.path.join(out, "n%d.err" % n)).read()))
        if str(n) not in ranks:
            problems.append("N=%d: RCCL's INIT log shows no %d-rank communicator (saw %s)" % (n, n, sorted(ranks)))
    runs.append({"n_gpus": n, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "scaling": d["scaling"],
                 "rccl_nranks": d["config"]["rccl_nranks"], "grad_sync": d["config"]["grad_sync"],
                 "capture_fallback": d["config"]["capture_fallback"], "line": d})
    print("[scale] N=%d: %.1f rays/s, %.4f ms/step, rccl_nranks %s, %s" % (n, d["value"], d["ms_per_step"], d["config"]["rccl_nranks"], d["config"]["grad_sync"]))
base = next((r["value"] for r in runs if r["n_gpus"] == 1), None)
for r in runs:
    r["weak_scaling_efficiency"] = round(r["value"] / (r["n_gpus"] * base), 4) if base else None
doc = {"skipped": not runs, "gpus_visible": ngpu, "metric": runs[0]["line"]["metric"] if runs else None,
       "sync_in_graph": os.environ.get("NERFHIP_SYNC_IN_GRAPH", "0") == "1", "runs": runs, "problems": problems}
if not runs:
    doc["reason"] = "no run produced a line (%d GPU(s) visible)" % ngpu
json.dump(doc, open(os.path.join(out, "SCALE.json"), "w"), indent=1)
print("[scale] wrote %s/SCALE.json: %d run(s), %d problem(s)" % (out, len(runs), len(problems)))
sys.exit(1 if problems else 0)
PY
