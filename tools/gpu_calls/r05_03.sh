# r05 call 3: per-tensor tables of the two new gradient tests (bounds are set from these)
set -u
OUT=gpurun_out/r05_03; mkdir -p $OUT
timeout 900 python -m pytest "tests/test_gpu_training.py::test_training_grads_fp32_all_48_tensors_in_full" "tests/test_gpu_bf16.py::test_timed_node_at_benchmark_size_vs_oracle_gradients" "tests/test_gpu_parity.py::test_fused_sample_pdf_indices_bit_exact" -q -m gpu -s > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt | tail -3
