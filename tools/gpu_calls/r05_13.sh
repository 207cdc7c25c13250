# r05 call 13: are the saving forward and the chain bound by their HBM writes?  timing-only builds whose activation / dY stores all
# land in 16 L2-resident tile blocks (results invalid), with and without the nt policy; shader clock / power while the step replays
set -u
OUT=gpurun_out/r05_13; mkdir -p $OUT
line() { python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], [(k["kernel"][:22], k["avg_launch_us"]) for k in d.get("roofline_kernels", [])[:4]])'; }
{
for i in 1 2; do
  echo "product build              $(python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 100 --warmup 10 2>/dev/null | line)"
  echo "stores wrap to 16 tiles nt $(NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_wrap16.so python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 100 --warmup 10 2>/dev/null | line)"
  echo "stores wrap to 16 tiles    $(NERFHIP_LIB_PATH=$PWD/nerf_pl_amd/variants/libnerfhip_wrap16pl.so python bench.py --no-cpu-baseline --no-extras --no-pmc --steps 100 --warmup 10 2>/dev/null | line)"
done
} 2>&1 | tee $OUT/exp_store_wrap.txt
( while true; do echo "t $(date +%s.%N)"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)"; sleep 0.25; done ) > $OUT/clocks_poll.txt 2>&1 &
POLL=$!
sleep 1
echo "bench start $(date +%s.%N)" > $OUT/clocks_marks.txt
python bench.py --steps 8000 --warmup 10 --no-extras --no-cpu-baseline --no-pmc > $OUT/long.json 2>/dev/null
echo "bench end $(date +%s.%N)" >> $OUT/clocks_marks.txt
sleep 1; kill $POLL
cat $OUT/clocks_marks.txt; grep -c sclk $OUT/clocks_poll.txt; grep -E "sclk|Power" $OUT/clocks_poll.txt | awk '{print $NF, $(NF-1)}' | paste - - | sort | uniq -c | sort -rn | head -20
