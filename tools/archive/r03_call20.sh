mkdir -p gpurun_out/r03u
OUT=gpurun_out/r03u
for rep in 1 2 3; do
for f in "" "--no-fuse-adam"; do
python bench.py --no-cpu-baseline --no-extras $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep$rep [$f]', d['ms_per_step'], d['value'])" | tee -a $OUT/adam_ab.txt
done
done
