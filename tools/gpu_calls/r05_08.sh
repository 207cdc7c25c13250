# r05 call 8: which change moved test_fp32_comparator_tracks_the_oracle_on_the_gate_scene (150-step trajectory, HIP fp32 vs oracle)?
set -u
OUT=gpurun_out/r05_08; mkdir -p $OUT
T="tests/test_gpu_psnr_gate.py::test_fp32_comparator_tracks_the_oracle_on_the_gate_scene"
( NERFHIP_ROW_TOTAL=exact timeout 400 python -m pytest $T -q -m gpu -s 2>&1 | grep -E "passed|failed|fp32 HIP vs oracle|AssertionError" | tail -4 ) | tee $OUT/exact_total.txt &
( NERFHIP_RENDER_FUSED=0 timeout 400 python -m pytest $T -q -m gpu -s 2>&1 | grep -E "passed|failed|fp32 HIP vs oracle|AssertionError" | tail -4 ) | tee $OUT/launches.txt &
( NERFHIP_RENDER_FUSED=0 NERFHIP_ROW_TOTAL=exact timeout 400 python -m pytest $T -q -m gpu -s 2>&1 | grep -E "passed|failed|fp32 HIP vs oracle|AssertionError" | tail -4 ) | tee $OUT/both_old.txt &
wait
