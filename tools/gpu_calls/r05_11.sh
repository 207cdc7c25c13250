# r05 call 11: sigma-head fold in the e4m3 and fp32 dW launches (tests + same-box A/B against the previous kernels), the bench's new
# setup / settle phase at the driver's settings, chain || dW overlap probe
set -u
OUT=gpurun_out/r05_11; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_fused_step.py tests/test_gpu_layered.py tests/test_gpu_render_fused.py -q -m gpu -k "not psnr" 2>&1 | tail -6 | tee $OUT/pytest_subset.txt
OLD=$PWD/nerf_pl_amd/variants/libnerfhip_oldbwd.so
line() { python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("cold_start_ms_per_step"), [(k["kernel"][:22], k["avg_launch_us"]) for k in d.get("roofline_kernels", [])[:4]])'; }
{
for i in 1 2; do
  echo "f8  fold     $(python bench.py --dtype bf16_f8 --no-cpu-baseline --no-extras --no-pmc --steps 100 --warmup 10 2>/dev/null | line)"
  echo "f8  previous $(NERFHIP_LIB_PATH=$OLD python bench.py --dtype bf16_f8 --no-cpu-baseline --no-extras --no-pmc --steps 100 --warmup 10 2>/dev/null | line)"
done
echo "fp32 fold     $(python bench.py --dtype fp32 --no-cpu-baseline --no-extras --no-pmc --steps 20 --warmup 5 2>/dev/null | line)"
echo "fp32 previous $(NERFHIP_LIB_PATH=$OLD python bench.py --dtype fp32 --no-cpu-baseline --no-extras --no-pmc --steps 20 --warmup 5 2>/dev/null | line)"
echo "bf16 (unchanged kernel) $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 100 --warmup 10 2>/dev/null | line)"
} 2>&1 | tee $OUT/ab_sigma_fold_f8_fp32.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_settings.json 2> $OUT/bench_driver_settings.err
python -c 'import json; d=json.loads(open("gpurun_out/r05_11/bench_driver_settings.json").read().strip().splitlines()[-1]); print("driver settings:", d["ms_per_step"], d.get("cold_start_ms_per_step"), d.get("setup"), d.get("f8_dw_ms_per_step"), d.get("non_mlp_us"))'
timeout 200 python tools/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/overlap_probe.txt
