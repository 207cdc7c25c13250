# r06 call 11: the folded final layer (f / dL/df not saved; dir job forms G; mlp_bwd_fold_kernel) — gradient tests first, then the step
set -u
OUT=gpurun_out/r06_11; mkdir -p $OUT
( time timeout 2400 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_step.py tests/test_gpu_bf16.py tests/test_gpu_render_fused.py tests/test_gpu_layered.py -q -m gpu --durations=5 2>&1 | tail -40 ) 2>&1 | tee $OUT/pytest_subset.txt
python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1500 $OUT/bench_default.json; tail -5 $OUT/bench_default.err
