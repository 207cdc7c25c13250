# r06 call 4: the whole -m gpu suite at this HEAD (PSNR gate vs the reference on a SNAPSHOT of the seeds minted so far), per-key parity errors
set -u
OUT=gpurun_out/r06_04; mkdir -p $OUT
cp tests/golden/reference_psnr_curves.json /tmp/curves_orig.json; cp gpurun_in/curves_merged_snapshot.json tests/golden/reference_psnr_curves.json
( time timeout 3000 python -m pytest tests -q -m gpu --durations=8 -s 2>&1 | grep -E "passed|failed|FAILED|Error|^[0-9.]+s |golden |render_rays 1024|coarse weights|PSNR vs reference|PSNR gate" | cut -c1-1800 ) 2>&1 | tee $OUT/pytest_gpu.txt
