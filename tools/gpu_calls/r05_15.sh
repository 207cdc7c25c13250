# r05 call 15: saved-tensor blocks interleaved over a workgroup's 8 tiles (NERFHIP_ACT_IL = 8) — tests of everything that writes or
# reads them, same-box A/B against the tile-major build (variants/libnerfhip_il1.so)
set -u
OUT=gpurun_out/r05_15; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_fused_step.py tests/test_gpu_render_fused.py tests/test_gpu_parity.py -q -m gpu -k "not psnr" -x 2>&1 | tail -8 | tee $OUT/pytest_subset.txt
line() { python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], [(k["kernel"][:22], k["avg_launch_us"]) for k in d.get("roofline_kernels", [])[:4]])'; }
{
for i in 1 2 3; do
  echo "interleaved (IL 8) $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 200 --warmup 10 2>/dev/null | line)"
  echo "tile-major  (IL 1) $(NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_il1.so python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 200 --warmup 10 2>/dev/null | line)"
done
} 2>&1 | tee $OUT/ab_interleave.txt
