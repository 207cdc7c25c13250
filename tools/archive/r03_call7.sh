mkdir -p gpurun_out/r03i
OUT=gpurun_out/r03i
# tensor-level A/B: chain looped (main) vs the round-2 chain (old): bf16 / fp32 storage must be bit-identical
NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_old.so python tests/tools/dbg_chain_ab.py --dump $OUT/dump_old.pt | tail -1
python tests/tools/dbg_chain_ab.py --dump $OUT/dump_main.pt | tail -1
NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_gate128.so python tests/tools/dbg_chain_ab.py --dump $OUT/dump_gate128.pt | tail -1
echo "=== old vs main"; python tests/tools/dbg_chain_ab.py --compare $OUT/dump_old.pt $OUT/dump_main.pt 2>&1 | tee $OUT/compare_old_main.txt
echo "=== main vs gate128"; python tests/tools/dbg_chain_ab.py --compare $OUT/dump_main.pt $OUT/dump_gate128.pt 2>&1 | grep -E "acts|flat" | tee $OUT/compare_main_gate128.txt
rm -f $OUT/dump_*.pt
for rep in 1 2; do
for v in main old gate128 prio; do
  if [ $v = main ]; then unset NERFHIP_LIB_PATH; else export NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_$v.so; fi
  for S in 192 64; do
    python tools/kbench.py --dtype bf16_f8 --samples $S --reps 20 2>/dev/null | tail -1 | sed "s/^/[$v rep$rep] /" | tee -a $OUT/kbench.txt
  done
done
done
unset NERFHIP_LIB_PATH
python -m pytest tests -q -m gpu --deselect tests/test_gpu_psnr_gate.py 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee $OUT/pytest.txt
