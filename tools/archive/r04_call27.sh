#!/bin/bash
mkdir -p gpurun_out/r04final3
timeout 280 python -m pytest tests -q -m gpu -k "not psnr_within and not comparator_tracks" > gpurun_out/r04final3/pytest_full.txt 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/r04final3/pytest_full.txt | tail -5 | tee gpurun_out/r04final3/pytest_gpu_no_psnr_gates.txt
