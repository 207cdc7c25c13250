#!/bin/bash
# A/B builds that differ ONLY in the backward-chain object: tools/build_chain_variant.sh <tag> [hipcc flags ...]
# -> nerf_pl_amd/variants/libnerfhip_<tag>.so = the main build's objects + mlp_bwd_chain.hip compiled with the given flags
TAG=$1; shift
cd "$(dirname "$0")/.." || exit 1
mkdir -p nerf_pl_amd/variants /tmp/chainv_$TAG
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" -c nerf_pl_amd/csrc/mlp_bwd_chain.hip -o /tmp/chainv_$TAG/mlp_bwd_chain.o || exit 1
OBJS=$(ls nerf_pl_amd/build/*.o | grep -v mlp_bwd_chain.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o nerf_pl_amd/variants/libnerfhip_$TAG.so $OBJS /tmp/chainv_$TAG/mlp_bwd_chain.o && echo "wrote nerf_pl_amd/variants/libnerfhip_$TAG.so"
