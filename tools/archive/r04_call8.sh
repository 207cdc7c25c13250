#!/bin/bash
OUT=gpurun_out/r04c8; mkdir -p $OUT
timeout 120 tools/probes/probe_mall.bin > $OUT/probe_mall.txt 2>&1; cat $OUT/probe_mall.txt
