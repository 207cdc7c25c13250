# r06 call 19: the reference leg of the PSNR gate on the grown fixture (30 minted reference seeds, 24 live)
set -u
OUT=gpurun_out/r06_19; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_psnr_gate.py -q -m gpu -s -k "reference" 2>&1 | grep -E "passed|failed|FAILED|Error|PSNR vs reference|^seed" | cut -c1-4000 ) 2>&1 | tee $OUT/pytest_psnr_ref.txt
