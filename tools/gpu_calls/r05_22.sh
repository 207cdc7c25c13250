# r05 call 22: the N > 1 step form at world 1 over RCCL — the two per-model all-reduces as one grouped launch vs one call each (ABAB)
set -u
OUT=gpurun_out/r05_22; mkdir -p $OUT
ms() { python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("launches_per_step"), d["config"]["grad_sync"][:40])'; }
{
for i in 1 2 3; do
  echo "grouped   $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 200 --warmup 10 --force-dist 2>/dev/null | ms)"
  echo "one each  $(NERFHIP_COALESCE_ALLREDUCE=0 python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 200 --warmup 10 --force-dist 2>/dev/null | ms)"
done
echo "no communicator $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 200 --warmup 10 2>/dev/null | ms)"
} | tee $OUT/ab_grouped_allreduce.txt
timeout 300 python -m pytest tests/test_bench_contract.py tests/test_gpu_fused_step.py -q -m gpu -k "rccl or dist or sync or torchrun" 2>&1 | tail -3
