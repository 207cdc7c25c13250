#!/bin/bash
# evidence set of the round's build + the read-pattern probe (does the dW kernel's read rate depend on the slab layout?)
mkdir -p gpurun_out/r04final
timeout 120 tools/probes/probe_read_pattern.bin > gpurun_out/r04final/probe_read_pattern.txt 2>&1; cat gpurun_out/r04final/probe_read_pattern.txt
bash tools/r04_final.sh r04final
