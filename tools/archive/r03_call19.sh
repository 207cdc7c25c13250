mkdir -p gpurun_out/r03s
OUT=gpurun_out/r03s
python -m pytest tests/test_gpu_layered.py -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee $OUT/pytest_layered.txt
python tools/layered_bench.py 2>&1 | grep -v "^\[\|amdgpu.ids" | tail -4 | tee $OUT/layered_bench.txt
