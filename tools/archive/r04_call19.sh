#!/bin/bash
# forward / chain: the two waves of a SIMD alternate their issue priority every n output tiles (prio1 / prio2 / prio4) vs the hardware's
# oldest-first arbitration (main)
OUT=gpurun_out/r04c19; mkdir -p $OUT
REPO=$(pwd)
for D in bf16 bf16_f8; do
  NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_prio2p.so timeout 120 python tools/stream_probe.py --dtype $D 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $OUT/stream_probe_prio2.txt
done
run() {  # tag, env...
  tag=$1; shift
  env "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
}
for rep in 1 2; do for v in main prio1 prio2 prio4; do
  L=$REPO/nerf_pl_amd/variants/libnerfhip_$v.so; [ $v = main ] && L=$REPO/nerf_pl_amd/libnerfhip.so
  run ${v}_$rep NERFHIP_LIB_PATH=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc
  [ $rep = 1 ] && run ${v}_f8 NERFHIP_LIB_PATH=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --no-pmc --dtype bf16_f8
  [ $rep = 1 ] && run ${v}_render NERFHIP_LIB_PATH=$L timeout 200 python bench.py --mode render --no-cpu-baseline
done; done
NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_prio2.so timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_fused_step.py tests/test_gpu_parity.py -q -x > $OUT/pytest_prio2.txt 2>&1; echo "tests(prio2) rc=$?"; tail -2 $OUT/pytest_prio2.txt
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=d.get('roofline_kernels') or []
    print(sys.argv[1].split('/')[-1].ljust(24), d['value'], d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')}, ' '.join('%s %.1f' % (k['kernel'].split('<')[0][4:]+('F' if 'fine pass' in k['kernel'] else 'C' if 'coarse pass' in k['kernel'] else ''), k['avg_launch_us']) for k in ks))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
