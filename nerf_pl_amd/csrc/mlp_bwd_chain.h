// launcher of the backward-chain kernels (mlp_bwd_chain.hip), called by the C ABI in mlp_bwd.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerfhip {
// per-model tensors of one chain launch (<= 2 models: workgroups [0, blocks0) run model 0, the rest model 1)
struct BwdChainArgs {
    const float* g_out[2];
    const float* out[2];
    int64_t n[2];
    const uint8_t* packed_bwd[2];
    const uint8_t* acts[2];
    uint8_t* dys[2];
    int blocks0;
};
// n_models (1 or 2) chains in ONE launch: g_out (n,4) [* g_scale], out (n,4), `tiles` = 32-point wave tiles of each model (a
// multiple of the workgroup's waves); all arrays hold n_models entries
void launch_bwd_chain(int n_models, const float* const* g_out, const float* g_scale, const float* const* out, const int64_t* n,
                      const void* const* packed_bwd, const void* const* acts, void* const* dys, int dtype, const int64_t* tiles,
                      hipStream_t s);
}  // namespace nerfhip
