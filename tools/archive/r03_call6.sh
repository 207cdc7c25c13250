mkdir -p gpurun_out/r03f
OUT=gpurun_out/r03f
python -m pytest tests -q -m gpu --deselect tests/test_gpu_psnr_gate.py -s 2>&1 | grep -vE "^RCCL|^HIP version|^ROCm|^Hostname|^Librccl" | tail -60 > $OUT/pytest_all.log
tail -40 $OUT/pytest_all.log
for rep in 1 2; do
for v in main old; do
  if [ $v = main ]; then unset NERFHIP_LIB_PATH; else export NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_$v.so; fi
  for S in 192 64; do
    python tools/kbench.py --dtype bf16_f8 --samples $S --reps 20 2>/dev/null | tail -1 | sed "s/^/[$v rep$rep] /" | tee -a $OUT/kbench.txt
  done
done
done
unset NERFHIP_LIB_PATH
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03f/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype','mlp_kernels_us_per_step','step_frac_mfma','bf16_storage_ms_per_step','render_fwd_rays_per_s_per_gpu','render_fwd_frac_mfma','eval_ms_per_image','eval_rays_per_s','eval_frac_mfma','traffic_note') if k in d})
for r in d['roofline_kernels']: print(r['kernel'], r['avg_launch_us'], r['frac_mfma'], r['frac_hbm'])
print(d['roofline_north_star']['avg_launch_us'], d['roofline_north_star']['frac'], d['cpu_baseline'])
PY
