# r06 call 18: W_c tiles by 4-wave workgroups (one memory round trip, K split over the waves) in 256-thread pack kernels: suites, ABAB, prologue / pack durations
set -u
OUT=gpurun_out/r06_18; mkdir -p $OUT
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_fused_step.py tests/test_gpu_bf16.py tests/test_gpu_render_fused.py tests/test_gpu_layered.py tests/test_gpu_inference.py tests/test_gpu_draws.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | cut -c1-400 ) 2>&1 | tee $OUT/pytest_subset.txt
for rep in 1 2 3; do
  for T in gpurun_in/bwdfold .; do
    ( cd $T && python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null ) | T=$T python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s' % ('HEAD (fwd+chain fold)' if os.environ['T']=='.' else 'before (febc0e3)'), 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], [(k['kernel'][:20], k['in_step_launch_us']) for k in d['roofline_kernels']], 'non-mlp', d['non_mlp_us'], 'north-star us', d['roofline_north_star']['avg_launch_us'])"
  done
done | tee $OUT/fwdfold_abab.txt
tools/ktrace_step.sh r06_18/trace --no-extras > $OUT/kernel_by_grid.txt
grep -i "prologue\|pack\|adam\|fold\|reduce" $OUT/trace/trace/t_kernel_stats.csv | cut -c1-60,150-260 | tee $OUT/small_kernels.txt
