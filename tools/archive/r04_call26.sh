#!/bin/bash
# final build: the default line (+ in-run PMC, extras, CPU baseline), the same command under rocprofv3 --kernel-trace --stats, the GPU suite without the two PSNR gates
TAG=r04final3; OUT=gpurun_out/$TAG; mkdir -p $OUT
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tools/ktrace_step.sh $TAG/trace > $OUT/kernel_by_grid.txt 2>&1
python bench.py --dtype bf16_f8 --no-cpu-baseline --no-extras > $OUT/bench_train_f8_dw.json 2>/dev/null
( time timeout 300 python -m pytest tests -q -m gpu -k "not psnr_within and not comparator_tracks" 2>&1 | tail -3 ) 2>&1 | tee $OUT/pytest_gpu_no_psnr_gates.txt
python - <<'PY'
import json
for f in ('bench_default','bench_train_f8_dw'):
    d=json.loads(open('gpurun_out/r04final3/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['dtype'], {k: d.get(k) for k in ('launches_per_step','non_mlp_us','mlp_kernels_us_per_step','step_frac_mfma')}, d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('traffic'))
PY
