mkdir -p gpurun_out/r03p
OUT=gpurun_out/r03p
for rep in 1 2; do
for W in 512 384 256; do
NERFHIP_DW_WGS=$W python tools/kbench.py --dtype bf16 --samples 192 --merged --reps 20 2>/dev/null | tail -1 | sed "s/^/[WGS=$W rep$rep] /" | tee -a $OUT/kbench_bf16_wgs.txt
done
done
for W in 512 256; do
NERFHIP_DW_WGS=$W python bench.py --dtype bf16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 WGS=$W', d['ms_per_step'], d['value'])" | tee -a $OUT/kbench_bf16_wgs.txt
done
NERFHIP_DW_WGS=256 python bench.py --dtype fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 WGS=256', d['ms_per_step'], d['value'])" | tee -a $OUT/kbench_bf16_wgs.txt
