# r06 call 14: the fold kernel on the fp32 MFMA; the new fold tests; same-box ABAB of the tree before the fold (fc229f5, built from source into
# gpurun_in/prefold) against HEAD through bench.py
set -u
OUT=gpurun_out/r06_14; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_step.py tests/test_bench_contract.py -q -m gpu -s 2>&1 | grep -E "passed|failed|FAILED|Error|fold vs float64" ) | tee $OUT/pytest_subset.txt
for rep in 1 2 3; do
  for T in gpurun_in/prefold .; do
    ( cd $T && python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null ) | T=$T python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-18s' % ('HEAD' if os.environ['T']=='.' else 'before (fc229f5)'), 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], 'launches', d['launches_per_step'], [(k['kernel'][:20], k['in_step_launch_us']) for k in d['roofline_kernels']], 'north-star us', d['roofline_north_star']['avg_launch_us'])"
  done
done | tee $OUT/fold_abab.txt
