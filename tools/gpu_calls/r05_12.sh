# r05 call 12: SQ issue accounts of the step's MLP kernels (tools/pmc_issue.sh), shader / memory clocks while the step replays
set -u
OUT=gpurun_out/r05_12; mkdir -p $OUT
bash tools/pmc_issue.sh r05_12/pmc bf16 > $OUT/pmc_issue.log 2>&1
cp $OUT/pmc/pmc_issue.txt $OUT/pmc_issue.txt 2>/dev/null; cat $OUT/pmc_issue.txt | cut -c1-400
( python bench.py --steps 6000 --warmup 10 --no-extras --no-cpu-baseline --no-pmc > $OUT/long.json 2>/dev/null & 
  BP=$!; sleep 12; for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" ; sleep 0.5; done; wait $BP ) > $OUT/clocks_during_step.txt 2>&1
head -30 $OUT/clocks_during_step.txt; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -5
