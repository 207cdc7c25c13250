#!/bin/bash
# dW ring A/B: main (byte-sized 160 KiB ring, no padded DMAs) vs noring (4-stage ring, no padded DMAs) vs dwpad (round 3: padded DMAs);
# + the upper bound of taking gate words / running maximum out of the saving forward (variant `small`, results invalid)
OUT=gpurun_out/r04c12; mkdir -p $OUT
REPO=$(pwd)
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_step.py tests/test_gpu_bf16.py -q -x > $OUT/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.txt
for rep in 1 2; do for v in main dwpad noring; do for D in bf16 bf16_f8; do
  if [ $v = main ]; then unset NERFHIP_LIB_PATH; else export NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_$v.so; fi
  python tools/kbench.py --merged --dtype $D --reps 20 2>/dev/null | tail -1 | tee -a $OUT/kbench_dw.txt
done; done; done
for v in main small; do for D in bf16 bf16_f8; do for S in 192 64; do
  if [ $v = main ]; then unset NERFHIP_LIB_PATH; else export NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_$v.so; fi
  python tools/kbench.py --dtype $D --samples $S --reps 20 2>/dev/null | tail -1 | tee -a $OUT/kbench_small_bound.txt
done; done; done
unset NERFHIP_LIB_PATH
python bench.py --no-extras --no-cpu-baseline > $OUT/bench_main.json 2> $OUT/bench_main.err
NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_dwpad.so python bench.py --no-extras --no-cpu-baseline > $OUT/bench_dwpad.json 2> $OUT/bench_dwpad.err
python bench.py --no-extras --no-cpu-baseline --dtype bf16_f8 > $OUT/bench_main_f8.json 2> $OUT/bench_main_f8.err
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('non_mlp_us', 'mlp_kernels_us_per_step', 'step_frac_mfma')})
    r=d['roofline']; print('   ', r['kernel'][:40], r['avg_launch_us'], r['frac'], r['traffic'])
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
