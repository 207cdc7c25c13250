#!/bin/bash
# dW (bf16): one software pipeline per tile (flat k-steps), 2-D wave grid; cycle accounts; cost model refit
OUT=gpurun_out/r04c16; mkdir -p $OUT
REPO=$(pwd)
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_fused_step.py -q -x > $OUT/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.txt
for v in dwprobe dwprobe1d; do
  export NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_$v.so
  echo "== $v, equal iterations" | tee -a $OUT/dw_probe.txt
  NERFHIP_DW_COST_A=1 NERFHIP_DW_COST_B=0 timeout 120 python tools/dw_probe.py 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $OUT/dw_probe.txt
done
echo "== dwprobe, cost-weighted plan (300, 35)" | tee -a $OUT/dw_probe.txt
timeout 120 python tools/dw_probe.py 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $OUT/dw_probe.txt
unset NERFHIP_LIB_PATH
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
}
for rep in 1 2; do
  run main_$rep X=1
  run main_eqiter_$rep NERFHIP_DW_COST_A=1 NERFHIP_DW_COST_B=0
  run main_a500_$rep NERFHIP_DW_COST_A=500 NERFHIP_DW_COST_B=35
  run v1d_$rep NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_v1d.so
  run v1drd4_$rep NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_v1drd4.so
  run dwpad_$rep NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_dwpad.so
done
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=[k for k in d['roofline_kernels'] if 'dw_kernel' in k['kernel']][0]
    print(sys.argv[1].split('/')[-1].ljust(28), d['ms_per_step'], {k: d.get(k) for k in ('non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')}, '| dW', r['avg_launch_us'], r['frac_hbm'])
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
