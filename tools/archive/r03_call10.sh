mkdir -p gpurun_out/r03l
OUT=gpurun_out/r03l
python -m pytest tests/test_gpu_layered.py -q -x 2>&1 | tail -15 | tee $OUT/pytest_layered.txt
python tools/layered_bench.py 2>&1 | grep -v "^\[" | tail -6 | tee $OUT/layered_bench.txt
python -m pytest tests -q -m gpu --deselect tests/test_gpu_psnr_gate.py --deselect tests/test_gpu_layered.py 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee $OUT/pytest.txt
