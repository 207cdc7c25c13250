#!/bin/bash
# dW kernel (bf16): compile-time job classes with a byte-sized 160 KiB ring (main) vs the same code with the 4-stage ring (ring4) vs
# round 3's kernel (dwpad); per-wave cycle accounts of the dW launch (dwprobe / dwprobe4)
OUT=gpurun_out/r04c13; mkdir -p $OUT
REPO=$(pwd)
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_fused_step.py -q -x > $OUT/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.txt
for v in dwprobe dwprobe4; do
  NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_$v.so timeout 120 python tools/dw_probe.py 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tee -a $OUT/dw_probe.txt
done
for rep in 1 2; do for v in main ring4 dwpad; do
  if [ $v = main ]; then unset NERFHIP_LIB_PATH; else export NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_$v.so; fi
  timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
done; done
unset NERFHIP_LIB_PATH
timeout 200 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_main_pmc.json 2> $OUT/bench_main_pmc.err
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d['roofline']
    print(sys.argv[1].split('/')[-1].ljust(24), d['ms_per_step'], {k: d.get(k) for k in ('non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')}, '| dW', r['avg_launch_us'], r['frac'], r['traffic'])
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
