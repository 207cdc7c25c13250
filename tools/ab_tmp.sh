L=$(python tools/kbench.py --dtype bf16_f8 2>/dev/null | cut -c1-200); echo "$L"
S=$(echo "$L" | sed 's/.*fwd+save \([0-9]*\).*/\1/')
python tools/kbench.py --dtype bf16 2>/dev/null | cut -c1-200
if [ "$S" -ge 290 ]; then echo "SLOWISH box"; fi
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_hunt.json
python -c "
import json; d=json.load(open('gpurun_out/bench_hunt.json')); print('step', d['ms_per_step'], d['value']); [print('  ', k['kernel'][:40], k['avg_launch_us']) for k in d['roofline_kernels'][:3]]"
