# r06 call 6: dW split-plan cost model, second sweep (towards bytes-proportional); row total exact vs aten in the step
set -u
OUT=gpurun_out/r06_06; mkdir -p $OUT
V=nerf_pl_amd/variants
for rep in 1 2 3; do
  for L in "" libnerfhip_dw_c150_45.so libnerfhip_dw_c100_50.so libnerfhip_dw_c50_55.so libnerfhip_dw_c0_60.so libnerfhip_dw_c150_45d4.so; do
    if [ -n "$L" ]; then export NERFHIP_LIB_PATH=$PWD/$V/$L; else unset NERFHIP_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ.get('NERFHIP_LIB_PATH','HEAD').split('/')[-1], d['ms_per_step'], [(k['kernel'][:18], k['in_step_launch_us'], k['avg_launch_us']) for k in d['roofline_kernels'][:1]])"
  done
done | tee $OUT/dw_plan_ab.txt
unset NERFHIP_LIB_PATH
for rep in 1 2; do
  for RT in aten exact; do
    export NERFHIP_ROW_TOTAL=$RT
    python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('row total', os.environ['NERFHIP_ROW_TOTAL'], d['ms_per_step'], [(k['kernel'][:18], k['in_step_launch_us']) for k in d['roofline_kernels']])"
  done
done | tee $OUT/row_total_ab.txt
