#!/bin/bash
OUT=gpurun_out/r04c7; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_draws.py tests/test_gpu_fused_step.py tests/test_gpu_training.py -q -x > $OUT/pytest.txt 2>&1; echo "tests rc=$?"; tail -6 $OUT/pytest.txt
for rep in 1 2; do for f in "" "--fuse-adam"; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline $f > $OUT/bench$rep$f.json 2> $OUT/bench$f.err
  python - "$OUT/bench$rep$f.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], {k:d.get(k) for k in ("ms_per_step","mlp_kernels_us_per_step","non_mlp_us","launches_per_step","step_frac_mfma")})
P
done; done
