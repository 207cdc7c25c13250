#!/bin/bash
# Multi-GPU scaling run of bench.py on ONE node (the driver's SCALE protocol): N = 1, 2, 4, 8 ranks, one per GPU, over
# RCCL/xGMI.  `python bench.py --gpus N` launches its own N ranks (torch.distributed.run) and refuses to print a line for any
# other world size; this script additionally checks the line (n_gpus == N, config.rccl_nranks == N) and RCCL's own INIT log.
#   tools/launch_scale.sh [outdir] [extra bench args...]
OUT=${1:-gpurun_out/scale}; shift || true
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=${NCCL_DEBUG:-INFO} NCCL_DEBUG_SUBSYS=INIT
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "[scale] only $NGPU GPU(s) visible: skipping N=$N"; continue; fi
  python bench.py --gpus $N --steps 50 --warmup 10 --no-cpu-baseline "$@" > $OUT/n$N.json 2> $OUT/n$N.err || { echo "[scale] N=$N failed (rc $?)"; tail -5 $OUT/n$N.err; continue; }
  python - "$OUT/n$N.json" "$OUT/n$N.err" $N <<'PY'
import json, re, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
n = int(sys.argv[3])
assert line, "no JSON line for N=%d" % n
d = json.loads(line[-1])
assert d["n_gpus"] == n, (d["n_gpus"], n)
if n > 1:
    assert d["config"]["rccl_nranks"] == n, d["config"]
    ranks = set(re.findall(r"nranks (\d+)", open(sys.argv[2]).read()))
    assert str(n) in ranks, "RCCL did not report a %d-rank communicator (saw %s)" % (n, sorted(ranks))
print("[scale] N=%d: %.1f rays/s, %.4f ms/step, rccl_nranks %s, %s" % (n, d["value"], d["ms_per_step"], d["config"]["rccl_nranks"], d["config"]["grad_sync"]))
PY
done
