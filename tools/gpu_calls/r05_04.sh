# r05 call 4: ATen-order row total as the default: parity + gradient tests again (per-tensor tables), the whole parity/draws files
set -u
OUT=gpurun_out/r05_04; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_draws.py "tests/test_gpu_training.py" "tests/test_gpu_bf16.py::test_timed_node_at_benchmark_size_vs_oracle_gradients" tests/test_gpu_fused_step.py tests/test_gpu_inference.py -q -m gpu -s > $OUT/pytest.txt 2>&1
grep -E "passed|failed|^FAILED" $OUT/pytest.txt | tail -12
