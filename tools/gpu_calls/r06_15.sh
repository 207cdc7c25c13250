# r06 call 15: xyz_encoding_final folded into the dir layer in the forward and in the chain (W_c = W_dir[:, :256] W_final formed by the pack
# kernels): the parity / gradient / step suites first
set -u
OUT=gpurun_out/r06_15; mkdir -p $OUT
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_fused_step.py tests/test_gpu_bf16.py tests/test_gpu_render_fused.py tests/test_gpu_layered.py tests/test_gpu_inference.py tests/test_gpu_draws.py -q -m gpu -s 2>&1 | grep -E "passed|failed|FAILED|Error|assert|fold vs float64" | cut -c1-400 ) 2>&1 | tee $OUT/pytest_subset.txt
python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], [(k['kernel'][:20], k['in_step_launch_us']) for k in d['roofline_kernels']], 'north-star us', d['roofline_north_star']['avg_launch_us'], 'non-mlp', d['non_mlp_us'])" | tee $OUT/bench_quick.txt
