#!/bin/bash
# bf16 dW: the sigma head's gradient formed by the final layer's workgroups (h8 read once) vs its own job (nofold)
OUT=gpurun_out/r04c25; mkdir -p $OUT
REPO=$(pwd)
timeout 400 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_training.py tests/test_gpu_fused_step.py -q -x -s -k "not psnr_at_equal" > $OUT/pytest.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|worst per-tensor" $OUT/pytest.txt | tail -6
run() {  # tag, env...
  tag=$1; shift
  env "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
}
run main_pmc X=1 timeout 300 python bench.py --no-cpu-baseline
run nofold_1 NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_nofold.so timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc
run main_1 X=1 timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=d.get('roofline_kernels') or []
    print(sys.argv[1].split('/')[-1].ljust(24), d['value'], d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')}, ' '.join('%s %.1f' % (k['kernel'].split('<')[0][4:]+('F' if 'fine pass' in k['kernel'] else 'C' if 'coarse pass' in k['kernel'] else ''), k['avg_launch_us']) for k in ks), [k.get('traffic') for k in ks][:1])
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
