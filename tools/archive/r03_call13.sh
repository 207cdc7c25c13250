mkdir -p gpurun_out/r03n
OUT=gpurun_out/r03n
for v in d0 main; do
  if [ $v = main ]; then unset NERFHIP_LIB_PATH; else export NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_$v.so; fi
  python tests/tools/dbg_chain_ab.py --dump $OUT/dump_$v.pt 2>/dev/null | tail -1
done
unset NERFHIP_LIB_PATH
echo "=== d0 vs main (D=2)"; python tests/tools/dbg_chain_ab.py --compare $OUT/dump_d0.pt $OUT/dump_main.pt 2>&1 | tail -12 | tee $OUT/compare.txt
rm -f $OUT/dump_*.pt
for rep in 1 2; do
for v in d0 d1 main d3; do
  if [ $v = main ]; then unset NERFHIP_LIB_PATH; else export NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_$v.so; fi
  for D in bf16_f8 bf16; do
    python tools/kbench.py --dtype $D --samples 192 --reps 20 2>/dev/null | tail -1 | sed "s/^/[$v rep$rep] /" | tee -a $OUT/kbench.txt
  done
  python tools/kbench.py --dtype bf16_f8 --samples 64 --reps 20 2>/dev/null | tail -1 | sed "s/^/[$v rep$rep] /" | tee -a $OUT/kbench.txt
done
done
