#!/bin/bash
OUT=gpurun_out/r04c18; mkdir -p $OUT
REPO=$(pwd)
for D in bf16 bf16_f8; do
  NERFHIP_LIB_PATH=$REPO/nerf_pl_amd/variants/libnerfhip_stprobe.so timeout 120 python tools/stream_probe.py --dtype $D 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $OUT/stream_probe.txt
done
