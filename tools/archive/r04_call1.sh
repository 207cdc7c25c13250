#!/bin/bash
# round 4, call 1: the fused launches against the ones they replace, the fused step, a first bf16 headline
OUT=gpurun_out/r04c1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_draws.py -x -q > $OUT/pytest_draws.txt 2>&1; echo "draws rc=$?"; tail -15 $OUT/pytest_draws.txt
timeout 900 python -m pytest tests/test_gpu_fused_step.py -q > $OUT/pytest_fused.txt 2>&1; echo "fused rc=$?"; tail -15 $OUT/pytest_fused.txt
timeout 300 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_bf16_noextras.json 2> $OUT/bench_bf16_noextras.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_bf16_noextras.json; tail -3 $OUT/bench_bf16_noextras.err
( time timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "bench default rc=$?"; cat $OUT/bench_default.time; tail -5 $OUT/bench_default.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r04c1/bench_default.json"))
for k in ("ms_per_step","dtype","mlp_kernels_us_per_step","non_mlp_us","launches_per_step","step_frac_mfma","f8_dw_ms_per_step","fp32_c1_ms_per_step","ndc_c3_ms_per_step","traffic_note","eval_ms_per_image"):
    print(k, d.get(k))
for r in d["roofline_kernels"]: print("  %-80s %7.1f us mfma %.3f hbm %.3f traffic %s"%(r["kernel"],r["avg_launch_us"],r["frac_mfma"],r["frac_hbm"],r["traffic"]))
P
