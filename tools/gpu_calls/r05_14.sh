# r05 call 14: the chip's write ceiling from compute kernels (the chain is bound by its 1.42 GB of dY stores: call 13)
set -u
OUT=gpurun_out/r05_14; mkdir -p $OUT
timeout 60 tools/probes/write_ceiling.bin 2>&1 | tee $OUT/write_ceiling.txt
timeout 60 tools/probes/write_pattern.bin 2>&1 | tee $OUT/write_pattern.txt
