python -m pytest tests -m gpu -q -s -k "f8 or gradient_direction" 2>&1 | grep "worst\|fp8-storage\|passed\|failed"
python tools/psnr_seeds.py --seeds 2 --dtypes bf16_f8 --out gpurun_out/r02m/psnr_seeds_dy_e5m2.json 2>&1 | grep -v amdgpu | head -4
