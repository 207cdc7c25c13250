#!/bin/bash
mkdir -p gpurun_out/r04c22
timeout 300 python -m pytest tests/test_gpu_bf16.py -q -s -k gradient_direction 2>&1 | grep -E "worst|passed|failed|Error|assert" | tee gpurun_out/r04c22/grad_rel.txt
