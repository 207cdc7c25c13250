# r06 call 5: the two tests that failed in call 4 (bounds fixed), dW split-plan cost-model variants, row total exact vs aten in the step
set -u
OUT=gpurun_out/r06_05; mkdir -p $OUT
cp gpurun_in/curves_merged_snapshot.json tests/golden/reference_psnr_curves.json
( timeout 900 python -m pytest "tests/test_gpu_parity.py::test_render_rays_fp32_benchmark_size_vs_oracle" "tests/test_gpu_psnr_gate.py::test_psnr_at_equal_steps_within_0p1_db_of_the_reference" tests/test_layout_host.py tests/test_gpu_fused_step.py -q -s 2>&1 | grep -E "passed|failed|render_rays 1024|coarse weights|PSNR vs reference:" | cut -c1-1500 ) | tee $OUT/pytest_two.txt
V=nerf_pl_amd/variants
for rep in 1 2; do
  for L in "" libnerfhip_dw_c200_40.so libnerfhip_dw_c400_30.so libnerfhip_dw_c150_45.so libnerfhip_dw_c300_28.so libnerfhip_dw_depth4.so; do
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
    NERFHIP_ROW_TOTAL=$RT python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('row total', os.environ['NERFHIP_ROW_TOTAL'], d['ms_per_step'], [(k['kernel'][:18], k['in_step_launch_us']) for k in d['roofline_kernels']])"
  done
done | tee $OUT/row_total_ab.txt
