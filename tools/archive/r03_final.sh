# Round-3 evidence set (run ON THE GPU BOX through gpurun): everything profiles/README.md "Round 3" quotes, one box, one call
set -u
OUT=gpurun_out/r03final; mkdir -p $OUT
REPO=$(pwd)
# PMC first: the default line below reports `traffic` only from counters stamped with THIS build's kernel-source digest
tools/pmc_kernels.sh r03final/pmc bf16_f8 > $OUT/pmc_traffic.txt 2>&1
[ -s gpurun_out/r03final/pmc/pmc_traffic.json ] && cp gpurun_out/r03final/pmc/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tools/ktrace_step.sh r03final/trace > $OUT/kernel_by_grid.txt 2>&1
python bench.py --dtype bf16 --no-cpu-baseline --no-extras > $OUT/bench_train_bf16_storage.json 2>/dev/null
python bench.py --dtype fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_train_fp32.json 2>/dev/null
python bench.py --mode render --no-cpu-baseline > $OUT/bench_render.json 2>/dev/null
python bench.py --mode eval --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_eval.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --modular-step > $OUT/bench_modular_step.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --fuse-adam > $OUT/bench_fuse_adam.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --force-dist --sync-in-graph 1 > $OUT/bench_rccl_world1_one_graph.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --force-dist --sync-in-graph 0 > $OUT/bench_rccl_world1_two_graphs.json 2>/dev/null
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['dtype'])
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_d0.so python tests/tools/dbg_chain_ab.py --dump $OUT/dump_d0.pt 2>/dev/null | tail -1
python tests/tools/dbg_chain_ab.py --dump $OUT/dump_main.pt 2>/dev/null | tail -1
python tests/tools/dbg_chain_ab.py --compare $OUT/dump_d0.pt $OUT/dump_main.pt > $OUT/chain_ring_tensor_ab.txt 2>&1; rm -f $OUT/dump_*.pt
python -m pytest tests -q -m gpu --deselect tests/test_gpu_psnr_gate.py 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee $OUT/pytest.txt
[ -n "${SKIP_PSNR:-}" ] || timeout 420 python tests/tools/psnr_gate.py --gate --out $OUT/psnr_gate.json > $OUT/psnr_gate.log 2>&1; grep -E '"mean"|"stderr"|fp32"' $OUT/psnr_gate.log | tail -8
ls $OUT
