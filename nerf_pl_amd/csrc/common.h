// Shared device helpers for libnerfhip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nerfhip.h"

#define NERFHIP_WAVE 64

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
