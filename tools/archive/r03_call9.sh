mkdir -p gpurun_out/r03k
OUT=gpurun_out/r03k
NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_old.so python tests/tools/dbg_chain_ab.py --dump $OUT/dump_old.pt | tail -1
python tests/tools/dbg_chain_ab.py --dump $OUT/dump_main.pt | tail -1
echo "=== old vs main"; python tests/tools/dbg_chain_ab.py --compare $OUT/dump_old.pt $OUT/dump_main.pt 2>&1 | grep -E "flat" | tee $OUT/compare_old_main.txt
rm -f $OUT/dump_*.pt
for rep in 1 2; do
for v in main old; do
  if [ $v = main ]; then unset NERFHIP_LIB_PATH; else export NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_$v.so; fi
  for D in bf16_f8 bf16; do
    python tools/kbench.py --dtype $D --samples 192 --reps 20 2>/dev/null | tail -1 | sed "s/^/[$v rep$rep] /" | tee -a $OUT/kbench.txt
  done
  python tools/kbench.py --dtype bf16_f8 --samples 64 --reps 20 2>/dev/null | tail -1 | sed "s/^/[$v rep$rep] /" | tee -a $OUT/kbench.txt
done
done
unset NERFHIP_LIB_PATH
python tools/kbench.py --dtype bf16_f8 --samples 192 --merged --reps 20 2>/dev/null | tail -1 | tee -a $OUT/kbench.txt
python tools/kbench.py --dtype fp32 --samples 192 --reps 5 2>/dev/null | tail -1 | tee -a $OUT/kbench.txt
NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_old.so python tools/kbench.py --dtype fp32 --samples 192 --reps 5 2>/dev/null | tail -1 | tee -a $OUT/kbench.txt
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['value'], d['mlp_kernels_us_per_step'], [(r['kernel'][:22], r['avg_launch_us']) for r in d['roofline_kernels']])"
python bench.py --no-cpu-baseline --no-extras --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['ms_per_step'], d['value'], d['mlp_kernels_us_per_step'], [(r['kernel'][:22], r['avg_launch_us']) for r in d['roofline_kernels']])"
python -m pytest tests -q -m gpu --deselect tests/test_gpu_psnr_gate.py 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee $OUT/pytest.txt
