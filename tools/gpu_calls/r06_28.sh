# r06 call 28: evidence refresh at the final build — the N>1 step forms at world 1 over RCCL, configs[3] by-grid trace, SQ issue accounts
set -u
OUT=gpurun_out/r06_28; mkdir -p $OUT
python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/dist_plain.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --no-pmc --force-dist > $OUT/dist_merged_one_graph.json 2>$OUT/dist_merged_one_graph.err
python bench.py --no-cpu-baseline --no-extras --no-pmc --force-dist --grad-sync-form per_model > $OUT/dist_per_model_one_graph.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --no-pmc --force-dist --sync-in-graph 0 > $OUT/dist_merged_two_graphs.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/dist_plain2.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-pmc --force-dist > $OUT/dist_torchrun_world1.json 2>/dev/null
for f in $OUT/dist_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[1].split('/')[-1].ljust(36), 'sustained', d['ms_per_step'], 'literal', d['literal_contract']['ms_per_step'], 'rccl_nranks', c.get('rccl_nranks'), 'grad_sync', str(c.get('grad_sync'))[:110], 'fallback', c.get('capture_fallback'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done | tee $OUT/dist_forms_world1.txt
tools/ktrace_step.sh r06_28/trace_c3 --no-extras --no-pmc --workload c3 > $OUT/kernel_by_grid_c3.txt 2>&1; tail -12 $OUT/kernel_by_grid_c3.txt
tail -c 1500 $OUT/trace_c3/bench_under_trace.json | head -c 600; echo
tools/pmc_issue.sh r06_28/pmc_issue bf16 > $OUT/pmc_issue.txt 2>&1; head -30 $OUT/pmc_issue.txt | cut -c1-260
