#!/bin/bash
# Multi-GPU scaling run of bench.py on ONE node (the driver's SCALE protocol): N = 1, 2, 4, 8 ranks, one per GPU, over
# RCCL/xGMI — ONE command for the first box that has more than one GPU:
#     tools/launch_scale.sh [outdir] [extra bench args...]        -> <outdir>/SCALE.json (+ per-run .json / .err / by-grid .csv files)
# `python bench.py --gpus N` launches its own N ranks (torch.distributed.run) and refuses to print a line for any other world
# size; this script additionally checks each line (n_gpus == N, config.rccl_nranks == N) and RCCL's own INIT log, and makes the
# first hardware run self-explaining:
#   * per N the DEFAULT step (one hipGraph, form=merged: the one-rank launches + ONE 4.77 MB all-reduce) -> n<N>.json, the line
#     SCALE.json's efficiency is computed from; rccl_nranks, capture_fallback and the grad_sync description are copied out of it;
#   * per N > 1 the two alternatives, each its own run: form=per_model (the fine model's all-reduce under the coarse model's
#     backward) -> n<N>_per_model.json, and the two-graph fallback form (--sync-in-graph 0) -> n<N>_two_graphs.json;
#   * per N a rocprofv3 --kernel-trace of 10 steps of the default form, summarised by (kernel, grid) over all ranks ->
#     n<N>_kernel_by_grid.csv (the RCCL kernel shows up there with its duration: the exposed wire time of the step).
OUT=${1:-gpurun_out/scale}; shift || true
mkdir -p $OUT
REPO=$(pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=${NCCL_DEBUG:-INFO} NCCL_DEBUG_SUBSYS=INIT
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
RAN=""
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "[scale] only $NGPU GPU(s) visible: skipping N=$N"; continue; fi
  python bench.py --gpus $N --steps 50 --warmup 10 --no-cpu-baseline "$@" > $OUT/n$N.json 2> $OUT/n$N.err || { echo "[scale] N=$N failed (rc $?)"; tail -5 $OUT/n$N.err; continue; }
  RAN="$RAN $N"
  if [ "$N" -gt 1 ]; then
    python bench.py --gpus $N --steps 50 --warmup 10 --no-cpu-baseline --no-extras --grad-sync-form per_model "$@" > $OUT/n${N}_per_model.json 2> $OUT/n${N}_per_model.err || echo "[scale] N=$N per_model failed"
    python bench.py --gpus $N --steps 50 --warmup 10 --no-cpu-baseline --no-extras --sync-in-graph 0 "$@" > $OUT/n${N}_two_graphs.json 2> $OUT/n${N}_two_graphs.err || echo "[scale] N=$N two_graphs failed"
  fi
  # kernel trace of the default form (every rank process writes its own trace: rocprofv3's environment is inherited through
  # torch.distributed.run); best effort — a failed trace never fails the scaling run
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $REPO/$OUT/trace_n$N -o t -- \
      python $REPO/bench.py --gpus $N --steps 10 --warmup 3 --settle 40 --no-cpu-baseline --no-extras "$@" > $REPO/$OUT/n${N}_under_trace.json 2> $REPO/$OUT/n${N}_trace.log ) || echo "[scale] N=$N trace failed"
  python - "$OUT/trace_n$N" "$OUT/n${N}_kernel_by_grid.csv" <<'PY'
import collections, csv, glob, sys
d = collections.defaultdict(list)
csv.field_size_limit(1 << 30)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("nerfhip::", "").split("(")[0][:60]
        d[(n, r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(sys.argv[2], "w") as fh:
    fh.write("kernel,grid,calls,avg_us,min_us,median_us,max_us\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        fh.write('"%s",%s,%d,%.1f,%.1f,%.1f,%.1f\n' % (k[0], k[1], len(v), sum(v) / len(v), v[0], v[len(v) // 2], v[-1]))
print("[scale] wrote %s (%d kernel/grid rows)" % (sys.argv[2], len(d)))
PY
  find $OUT/trace_n$N -name "*.csv" -size +2M -delete 2>/dev/null
done
python - "$OUT" "$NGPU" $RAN <<'PY'
import json, os, re, sys
out, ngpu, ns = sys.argv[1], int(sys.argv[2]), [int(x) for x in sys.argv[3:]]
runs, problems = [], []


def last_line(path):
    if not os.path.exists(path):
        return None
    line = [l for l in open(path) if l.startswith("{")]
    return json.loads(line[-1]) if line else None


def brief(d):
    return None if d is None else {"value": d["value"], "ms_per_step": d["ms_per_step"], "rccl_nranks": d["config"]["rccl_nranks"],
                                   "grad_sync": d["config"]["grad_sync"], "capture_fallback": d["config"]["capture_fallback"],
                                   "launches_per_step": d.get("launches_per_step"), "literal_contract": d.get("literal_contract")}


for n in ns:
    d = last_line(os.path.join(out, "n%d.json" % n))
    if d is None:
        problems.append("no JSON line for N=%d" % n)
        continue
    if d["n_gpus"] != n:
        problems.append("N=%d: line says n_gpus %s" % (n, d["n_gpus"]))
    if n > 1:
        if d["config"]["rccl_nranks"] != n:
            problems.append("N=%d: rccl_nranks %s" % (n, d["config"]["rccl_nranks"]))
        ranks = set(re.findall(r"nranks (\d+)", open(os.path.join(out, "n%d.err" % n)).read()))
        if str(n) not in ranks:
            problems.append("N=%d: RCCL's INIT log shows no %d-rank communicator (saw %s)" % (n, n, sorted(ranks)))
        if d["config"]["capture_fallback"]:
            problems.append("N=%d: the one-graph capture fell back to two graphs: %s" % (n, d["config"]["capture_fallback"]))
    by_grid = os.path.join(out, "n%d_kernel_by_grid.csv" % n)
    runs.append({"n_gpus": n, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "scaling": d["scaling"],
                 "rccl_nranks": d["config"]["rccl_nranks"], "grad_sync": d["config"]["grad_sync"],
                 "capture_fallback": d["config"]["capture_fallback"], "literal_contract": d.get("literal_contract"),
                 "alternatives": {"per_model": brief(last_line(os.path.join(out, "n%d_per_model.json" % n))),
                                  "two_graphs": brief(last_line(os.path.join(out, "n%d_two_graphs.json" % n)))} if n > 1 else None,
                 "kernel_by_grid": by_grid if os.path.exists(by_grid) else None, "line": d})
    print("[scale] N=%d: %.1f rays/s, %.4f ms/step, rccl_nranks %s, capture_fallback %s, %s" % (
        n, d["value"], d["ms_per_step"], d["config"]["rccl_nranks"], d["config"]["capture_fallback"], d["config"]["grad_sync"]))
    for k, v in ((runs[-1]["alternatives"] or {}).items()):
        if v is not None:
            print("[scale]        %s: %.4f ms/step (%s)" % (k, v["ms_per_step"], v["grad_sync"]))
base = next((r["value"] for r in runs if r["n_gpus"] == 1), None)
for r in runs:
    r["weak_scaling_efficiency"] = round(r["value"] / (r["n_gpus"] * base), 4) if base else None
doc = {"skipped": not runs, "gpus_visible": ngpu, "metric": runs[0]["line"]["metric"] if runs else None,
       "default_form": "one hipGraph, GradSync form=merged", "runs": runs, "problems": problems}
if not runs:
    doc["reason"] = "no run produced a line (%d GPU(s) visible)" % ngpu
json.dump(doc, open(os.path.join(out, "SCALE.json"), "w"), indent=1)
print("[scale] wrote %s/SCALE.json: %d run(s), %d problem(s)" % (out, len(runs), len(problems)))
sys.exit(1 if problems else 0)
PY
