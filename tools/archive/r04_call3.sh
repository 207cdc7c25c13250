#!/bin/bash
OUT=gpurun_out/r04c3; mkdir -p $OUT
python tools/probes/probe_randn.py > $OUT/probe_randn.txt 2>&1; cat $OUT/probe_randn.txt | tail -40
timeout 600 python -m pytest tests/test_gpu_draws.py -q > $OUT/pytest_draws.txt 2>&1; echo "draws rc=$?"; grep -E "passed|failed|FAILED" $OUT/pytest_draws.txt | tail -30
timeout 300 python tools/small_kernel_bench.py > $OUT/small.txt 2>&1; tail -20 $OUT/small.txt
