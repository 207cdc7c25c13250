#!/bin/bash
# dW (bf16): cost-weighted split plan vs equal iterations; DMA padding / stage pitch A/B of the class-based kernel; cycle accounts
OUT=gpurun_out/r04c14; mkdir -p $OUT
REPO=$(pwd)
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_fused_step.py -q -x > $OUT/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.txt
export NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_dwprobe.so
echo "== cost-weighted plan" | tee -a $OUT/dw_probe.txt
timeout 120 python tools/dw_probe.py 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $OUT/dw_probe.txt
echo "== equal iterations (NERFHIP_DW_COST_A=1 NERFHIP_DW_COST_B=0)" | tee -a $OUT/dw_probe.txt
NERFHIP_DW_COST_A=1 NERFHIP_DW_COST_B=0 timeout 120 python tools/dw_probe.py 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $OUT/dw_probe.txt
unset NERFHIP_LIB_PATH
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
}
for rep in 1 2; do
  run main_$rep
  run main_eqiter_$rep NERFHIP_DW_COST_A=1 NERFHIP_DW_COST_B=0
  run ring4_$rep NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_ring4.so
  run lpw5_$rep NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_lpw5.so
  run stage36_$rep NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_stage36.so
  run old36_$rep NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_old36.so
  run dwpad_$rep NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_dwpad.so
done
timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc --dtype bf16_f8 > $OUT/bench_f8_main.json 2> $OUT/bench_f8_main.err
NERFHIP_DW_COST_A=1 NERFHIP_DW_COST_B=0 timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc --dtype bf16_f8 > $OUT/bench_f8_eqiter.json 2> $OUT/bench_f8_eqiter.err
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=[k for k in d['roofline_kernels'] if 'dw_kernel' in k['kernel']][0]
    print(sys.argv[1].split('/')[-1].ljust(28), d['ms_per_step'], {k: d.get(k) for k in ('non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')}, '| dW', r['avg_launch_us'], r['frac_hbm'])
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
