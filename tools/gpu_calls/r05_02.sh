# r05 call 2: the parity additions of this round on hardware
set -u
OUT=gpurun_out/r05_02; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_draws.py tests/test_gpu_training.py tests/test_gpu_fused_step.py "tests/test_gpu_bf16.py::test_timed_node_at_benchmark_size_vs_oracle_gradients" "tests/test_gpu_psnr_gate.py::test_psnr_gate_at_the_headline_sampling_64_plus_128" -q -m gpu -s --durations=6 > $OUT/pytest.txt 2>&1
grep -E "passed|failed|FAILED|Error|worst|gr3|timed node|PSNR gate|fused sample_pdf|^[0-9.]+s " $OUT/pytest.txt | tail -40
