#!/bin/bash
OUT=gpurun_out/r04c6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_draws.py tests/test_gpu_fused_step.py tests/test_gpu_training.py tests/test_gpu_parity.py tests/test_gpu_inference.py -q -x > $OUT/pytest.txt 2>&1; echo "tests rc=$?"; tail -6 $OUT/pytest.txt
timeout 200 python tools/small_kernel_bench.py 2>&1 | tail -2
for f in "" "--fuse-adam"; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline $f > $OUT/bench$f.json 2> $OUT/bench$f.err
  python - "$OUT/bench$f.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], {k:d.get(k) for k in ("ms_per_step","mlp_kernels_us_per_step","non_mlp_us","launches_per_step","step_frac_mfma")})
P
done
for f in "" "--fuse-adam"; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline $f > $OUT/bench2$f.json 2> /dev/null
  python - "$OUT/bench2$f.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], {k:d.get(k) for k in ("ms_per_step","mlp_kernels_us_per_step","non_mlp_us","launches_per_step")})
P
done
