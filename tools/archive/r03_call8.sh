mkdir -p gpurun_out/r03j
OUT=gpurun_out/r03j
NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_old.so python tests/tools/dbg_chain_ab.py --dump $OUT/dump_old.pt | tail -1
python tests/tools/dbg_chain_ab.py --dump $OUT/dump_main.pt | tail -1
echo "=== old vs main"; python tests/tools/dbg_chain_ab.py --compare $OUT/dump_old.pt $OUT/dump_main.pt 2>&1 | grep -E "dys|flat" | tee $OUT/compare_old_main.txt
rm -f $OUT/dump_*.pt
python -m pytest tests -q -m gpu --deselect tests/test_gpu_psnr_gate.py 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee $OUT/pytest.txt
