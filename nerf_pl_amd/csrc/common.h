// Shared device helpers for libnerfhip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nerfhip.h"

// hipcc defaults to -ffp-contract=fast for device code.  The reference computes every mul/add as a
// separately rounded fp32 ATen op, so all arithmetic in this library is compiled with contraction
// OFF and written with the plain operators below (HIP's __fadd_rn/__fmul_rn wrappers are parsed
// inside HIP's headers with contraction still on, and fuse again after inlining).
#pragma clang fp contract(off)

#define NERFHIP_WAVE 64

__device__ __forceinline__ float nh_add(float a, float b) { return a + b; }
__device__ __forceinline__ float nh_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float nh_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float nh_div(float a, float b) { return a / b; }   // IEEE-correct (hipcc default)

#define NERFHIP_CHECK_ARG(cond) \
    do {                        \
        if (!(cond)) return NERFHIP_E_BADARG; \
    } while (0)

static inline int nerfhip_launch_status() { return (int)hipGetLastError(); }

namespace nerfhip {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- wave-wide reductions / scans (64 lanes, no LDS) ---------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix product / sum over the 64 lanes (fp64: tracks torch-CPU cumprod/cumsum,
// which accumulate fp32 inputs in fp64 and round per element — SURVEY A.9).
__device__ __forceinline__ double wave_incl_prod(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}
__device__ __forceinline__ double wave_incl_sum(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

}  // namespace nerfhip

#ifndef NERFHIP_STREAM_PROBE
#define NERFHIP_STREAM_PROBE 0     // debug builds: the activation-saving forward / the backward chain record the time their waves spend at
#endif                             // the weight ring's s_waitcnt and s_barrier (results of the launch are invalid; tools/stream_probe.py)
