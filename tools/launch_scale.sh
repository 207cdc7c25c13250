#!/bin/bash
# Multi-GPU scaling run of bench.py on ONE node (the driver's SCALE protocol): N = 1, 2, 4, 8 ranks, one per GPU, over
# RCCL/xGMI.  Checks that the JSON line reports n_gpus == N and that RCCL came up with N ranks.
#   tools/launch_scale.sh [outdir] [extra bench args...]
OUT=${1:-gpurun_out/scale}; shift || true
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=${NCCL_DEBUG:-INFO} NCCL_DEBUG_SUBSYS=INIT
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "[scale] only $NGPU GPU(s) visible: skipping N=$N"; continue; fi
  PORT=$((29500 + N))
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline "$@" > $OUT/n$N.json 2> $OUT/n$N.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --steps 50 --warmup 10 "$@" > $OUT/n$N.json 2> $OUT/n$N.err
  fi
  python - "$OUT/n$N.json" "$OUT/n$N.err" $N <<'PY'
import json, re, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
n = int(sys.argv[3])
assert line, "no JSON line for N=%d" % n
d = json.loads(line[-1])
assert d["n_gpus"] == n, (d["n_gpus"], n)
ranks = set(re.findall(r"nranks (\d+)", open(sys.argv[2]).read()))
if n > 1:
    assert str(n) in ranks, "RCCL did not report a %d-rank communicator (saw %s)" % (n, sorted(ranks))
print("[scale] N=%d: %.1f rays/s, %.4f ms/step, RCCL nranks %s" % (n, d["value"], d["ms_per_step"], sorted(ranks) or "-"))
PY
done
