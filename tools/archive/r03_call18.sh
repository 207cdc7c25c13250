mkdir -p gpurun_out/r03r
OUT=gpurun_out/r03r
for W in 256 512 384 256; do
NERFHIP_DW_WGS=$W python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f8 WGS=$W', d['ms_per_step'], d['value'], d['mlp_kernels_us_per_step'], [r['avg_launch_us'] for r in d['roofline_kernels']])" | tee -a $OUT/wgs.txt
done
for W in 512 256 512 256; do
NERFHIP_DW_WGS=$W python bench.py --dtype fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 WGS=$W', d['ms_per_step'], d['value'])" | tee -a $OUT/wgs.txt
done
