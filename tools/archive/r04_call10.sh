#!/bin/bash
SKIP_TESTS=1 bash tools/r04_final.sh r04final
timeout 400 python -m pytest tests/test_gpu_psnr_gate.py -q -k oracle 2>&1 | grep -E "passed|failed|HIP vs oracle" | tee gpurun_out/r04final/pytest_oracle_leg.txt
