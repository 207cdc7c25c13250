// launcher of the backward-chain kernels (mlp_bwd_chain.hip), called by the C ABI in mlp_bwd.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerfhip {
// one model: g_out (n,4) [* g_scale], out (n,4), `tiles` = 32-point wave tiles (a multiple of the workgroup's waves)
void launch_bwd_chain(const float* g_out, const float* g_scale, const float* out, int64_t n, const void* packed_bwd,
                      const void* acts, void* dys, int dtype, int64_t tiles, hipStream_t s);
}  // namespace nerfhip
