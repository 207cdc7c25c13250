# r06 call 24: does the dW kernel with the (default-off) regenerating job classes compiled in run the default step as fast as HEAD's?
# same-box ABAB: HEAD tree (2f744db kernels, gpurun_in/head) vs this tree with NERFHIP_REGEN_ENC=0 (default) and =1
set -u
OUT=gpurun_out/r06_24; mkdir -p $OUT
for rep in 1 2 3 4; do
  for V in head off on; do
    T=.; R=0; if [ $V = head ]; then T=gpurun_in/head; fi; if [ $V = on ]; then R=1; fi
    ( cd $T && NERFHIP_REGEN_ENC=$R python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null ) | V=$V python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-5s' % os.environ['V'], 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], [(k['kernel'][:20], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels']], 'non-mlp', d['non_mlp_us'])"
  done
done | tee $OUT/abab.txt
