#!/bin/bash
# (1) the final dW build: parity tests + bench with in-run PMC; (2) where the saving forward / chain waves wait (stream probe)
OUT=gpurun_out/r04c17; mkdir -p $OUT
REPO=$(pwd)
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_fused_step.py tests/test_gpu_layered.py -q -x > $OUT/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.txt
for D in bf16 bf16_f8; do
  NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_stprobe.so timeout 120 python tools/stream_probe.py --dtype $D 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $OUT/stream_probe.txt
done
timeout 300 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_main_pmc.json 2> $OUT/bench_main_pmc.err
timeout 300 python bench.py --no-extras --no-cpu-baseline --dtype bf16_f8 --no-pmc > $OUT/bench_f8.json 2> $OUT/bench_f8.err
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1].ljust(28), d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')})
    for r in d['roofline_kernels']: print('    %-70s %7.1f us  hbm %.3f mfma %.3f traffic %s alg %s' % (r['kernel'][:70], r['avg_launch_us'], r['frac_hbm'], r['frac_mfma'], r.get('traffic'), r['hbm_bytes']))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
