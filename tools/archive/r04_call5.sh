#!/bin/bash
OUT=gpurun_out/r04c5; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > $OUT/pytest_gpu.txt 2>&1 ) 2> $OUT/pytest_gpu.time; echo "gpu tests rc=$?"; tail -25 $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.time
( time timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "bench default rc=$?"; cat $OUT/bench_default.time; tail -5 $OUT/bench_default.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r04c5/bench_default.json"))
for k in ("ms_per_step","dtype","mlp_kernels_us_per_step","non_mlp_us","launches_per_step","step_frac_mfma","f8_dw_ms_per_step","fp32_c1_ms_per_step","ndc_c3_ms_per_step","eval_ms_per_image"):
    print(k, d.get(k))
for r in d["roofline_kernels"]: print("  %-80s %7.1f us mfma %.3f hbm %.3f traffic %s"%(r["kernel"],r["avg_launch_us"],r["frac_mfma"],r["frac_hbm"],r["traffic"]))
P
timeout 200 python tools/small_kernel_bench.py 2>&1 | tail -3
