#!/bin/bash
OUT=gpurun_out/r04c4; mkdir -p $OUT
python tools/probes/probe_randn_dump.py $OUT/randn_dump.npz 2>&1 | tail -3
