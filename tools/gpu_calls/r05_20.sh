# r05 call 20: Adam inside the reduce (5 launches) vs its own launch (6), ABBA x 2, 300 steps each, one more box
set -u
OUT=gpurun_out/r05_20; mkdir -p $OUT
ms() { python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("launches_per_step"))'; }
{
for i in 1 2; do
  echo "separate Adam   $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 300 --warmup 20 2>/dev/null | ms)"
  echo "Adam in reduce  $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 300 --warmup 20 --fuse-adam 2>/dev/null | ms)"
  echo "Adam in reduce  $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 300 --warmup 20 --fuse-adam 2>/dev/null | ms)"
  echo "separate Adam   $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 300 --warmup 20 2>/dev/null | ms)"
done
} | tee $OUT/ab_adam_in_reduce.txt
