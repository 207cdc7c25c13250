#!/bin/bash
# bf16 dW: measured per-class iteration costs in the split plan (default) vs the linear model (300 + 35/KiB) vs equal iterations;
# e4m3 dW: cycle accounts under equal iterations
OUT=gpurun_out/r04c20; mkdir -p $OUT
REPO=$(pwd)
NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_dwprobe.so timeout 120 python tools/dw_probe.py --dtype bf16_f8 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $OUT/dw_probe_f8.txt
NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_dwprobe.so timeout 120 python tools/dw_probe.py --dtype bf16 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $OUT/dw_probe_bf16_table_plan.txt
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
}
for rep in 1 2 3; do
  run table_$rep X=1
  run linear_$rep NERFHIP_DW_COST_A=300 NERFHIP_DW_COST_B=35
done
run eqiter_1 NERFHIP_DW_COST_A=1 NERFHIP_DW_COST_B=0
timeout 300 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_fused_step.py -q -x > $OUT/pytest.txt 2>&1; echo "tests rc=$?"; tail -2 $OUT/pytest.txt
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=d.get('roofline_kernels') or []
    print(sys.argv[1].split('/')[-1].ljust(24), d['value'], d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')}, ' '.join('%s %.1f' % (k['kernel'].split('<')[0][4:]+('F' if 'fine pass' in k['kernel'] else 'C' if 'coarse pass' in k['kernel'] else ''), k['avg_launch_us']) for k in ks))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
