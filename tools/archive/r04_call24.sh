#!/bin/bash
# bf16 dW: a stage's last REM pieces shared between the waves under EXEC masks (main) vs re-fetched by the surplus waves (noshare)
OUT=gpurun_out/r04c24; mkdir -p $OUT
REPO=$(pwd)
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_fused_step.py -q -x > $OUT/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.txt
run() {  # tag, env...
  tag=$1; shift
  env "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
}
for rep in 1 2; do
  run main_$rep X=1 timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc
  run noshare_$rep NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_noshare.so timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc
done
run main_pmc X=1 timeout 300 python bench.py --no-cpu-baseline
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=d.get('roofline_kernels') or []
    print(sys.argv[1].split('/')[-1].ljust(24), d['value'], d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')}, ' '.join('%s %.1f' % (k['kernel'].split('<')[0][4:]+('F' if 'fine pass' in k['kernel'] else 'C' if 'coarse pass' in k['kernel'] else ''), k['avg_launch_us']) for k in ks), [k.get('traffic') for k in ks][:1])
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
