python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for d in bf16 bf16_f8; do python tools/kbench.py --dtype $d 2>/dev/null | cut -c1-190; NERFHIP_LIB_PATH=nerf_pl_amd/variants/libnerfhip_sm.so python tools/kbench.py --dtype $d 2>/dev/null | cut -c1-190; done
python tools/kbench.py --dtype bf16_f8 --samples 64 2>/dev/null | cut -c1-190; NERFHIP_LIB_PATH=nerf_pl_amd/variants/libnerfhip_sm.so python tools/kbench.py --dtype bf16_f8 --samples 64 2>/dev/null | cut -c1-190
python tools/psnr_vs_oracle.py --steps 250 --every 25 --out gpurun_out/r02p/psnr_vs_oracle.json 2>&1 | grep -v amdgpu | tail -30
