mkdir -p gpurun_out/r03e
OUT=gpurun_out/r03e
export NERFHIP_DW_NOGROUP=1
for v in c_base c_pk c_sa c_sched; do
  NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_$v.so python tests/tools/dbg_chain_ab.py --dump $OUT/dump_$v.pt 2>&1 | tail -1
done
for v in c_pk c_sa c_sched; do echo "=== c_base vs $v"; python tests/tools/dbg_chain_ab.py --compare $OUT/dump_c_base.pt $OUT/dump_$v.pt; done 2>&1 | tee $OUT/compare.txt
unset NERFHIP_DW_NOGROUP
# dW grouping (main lib's mlp_bwd.o is linked into every c_* variant): grouped vs ungrouped with the SAME chain
NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_c_base.so python tests/tools/dbg_chain_ab.py --dump $OUT/dump_c_base_grouped.pt | tail -1
echo "=== dW ungrouped vs grouped (c_base chain)"; python tests/tools/dbg_chain_ab.py --compare $OUT/dump_c_base.pt $OUT/dump_c_base_grouped.pt 2>&1 | grep -E "flat" | tee -a $OUT/compare.txt
for v in c_base; do
  for g in 1 0; do
    if [ $g = 1 ]; then unset NERFHIP_DW_NOGROUP; else export NERFHIP_DW_NOGROUP=1; fi
    NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_$v.so python tools/kbench.py --dtype bf16_f8 --samples 192 --reps 20 2>/dev/null | tail -1 | sed "s/^/[group=$g] /" | tee -a $OUT/kbench.txt
    NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_$v.so python tools/kbench.py --dtype bf16_f8 --samples 64 --reps 20 2>/dev/null | tail -1 | sed "s/^/[group=$g] /" | tee -a $OUT/kbench.txt
  done
done
rm -f $OUT/dump_*.pt
