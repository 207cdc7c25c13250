# r05 call 6: same-box A/B of the single-launch forward (on / off), Adam inside the reduce (5 launches) or its own launch (6),
# test_time renders through the single-launch kernel or not; trace of the step
set -u
OUT=gpurun_out/r05_06; mkdir -p $OUT
run() { tag=$1; shift; env "$@" > /dev/null 2>&1; }
b() { tag=$1; shift; "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2].ljust(28), d['value'], d['ms_per_step'], {k: d.get(k) for k in ('launches_per_step','non_mlp_us','mlp_kernels_us_per_step','step_frac_mfma','eval_ms_per_image','render_fwd_rays_per_s_per_gpu')})
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
A="--no-cpu-baseline --no-extras --no-pmc"
b one_launch_fwd           python bench.py $A
b four_launch_fwd          env NERFHIP_RENDER_FUSED=0 python bench.py $A
b one_launch_fwd_fuse_adam python bench.py $A --fuse-adam
b one_launch_fwd_2         python bench.py $A
b four_launch_fwd_2        env NERFHIP_RENDER_FUSED=0 python bench.py $A
b fuse_adam_2              python bench.py $A --fuse-adam
b f8_one_launch            python bench.py $A --dtype bf16_f8
b f8_four_launch           env NERFHIP_RENDER_FUSED=0 python bench.py $A --dtype bf16_f8
b eval_launches            python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline
b eval_one_launch          env NERFHIP_FUSE_TEST_TIME=1 python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline
b render_one_launch        python bench.py --mode render --no-cpu-baseline
b render_launches          env NERFHIP_RENDER_FUSED=0 python bench.py --mode render --no-cpu-baseline
tools/ktrace_step.sh r05_06/trace > $OUT/kernel_by_grid.txt 2>&1; head -14 $OUT/trace/kernel_by_grid.csv
timeout 300 python -m pytest tests/test_gpu_training.py::test_fine_pass_grads_fp32_on_identical_depths -q -m gpu -s 2>&1 | grep -E "passed|failed|fine pass|^E " | tail -5
