mkdir -p gpurun_out/r03l
OUT=gpurun_out/r03l
python -m pytest tests/test_gpu_layered.py -q 2>&1 | tail -5 | tee $OUT/pytest_layered.txt
python tools/layered_bench.py 2>&1 | grep -v "^\[" | tail -6 | tee $OUT/layered_bench.txt
NERFHIP_LIN_TJ=128 python tools/layered_bench.py 2>&1 | grep -v "^\[" | tail -6 | sed 's/^/[TJ128] /' | tee -a $OUT/layered_bench.txt
