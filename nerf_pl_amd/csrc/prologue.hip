// The prologue of a training step in ONE launch: everything that depends only on the generator state and on the parameters.
//   * the batch and the step's random draws      (draws.hip: nerfhip_torch_draws with a ray batch — train.py:89-94, rendering.py:203,
//                                                 :152, :39, :152)
//   * both models' packed weight images          (mlp_pack.hip: nerfhip_mlp_pack_weights_train_multi — forward stream + W^T stream)
// Two independent jobs that each used to be a graph node of their own at the head of the step; the workgroups [0, draw_blocks)
// run the draws, the rest pack one 1 KiB piece per wave.  Same device code as the two launches (draws_body.h,
// mlp_pack_pieces.h): same bits.
#include "draws_body.h"
#include "mlp_pack_pieces.h"

namespace nerfhip {

template <int PREC>
__global__ __launch_bounds__(256) void train_prologue_kernel(DrawTable T, unsigned long long seed_v, unsigned long long offset_v,
                                                             unsigned long long* __restrict__ state, int draw_blocks,
                                                             MultiPackTable P, int n_models) {
    if ((int)blockIdx.x < draw_blocks) {
        philox_draws_block(T, seed_v, offset_v, state, (int)blockIdx.x, draw_blocks);
        return;
    }
    // the rest of the grid packs: pack_blocks() workgroups per model (mlp_pack_pieces.h pack_model_block: the W_c tiles of the folded
    // layer first, then four 1 KiB pieces per workgroup).  The model index is uniform over the workgroup: its tables are read straight
    // from the kernel arguments with scalar loads.  (Round 4 first selected them into a local ParamTable; pack_*_piece index that
    // table by layer, a dynamically indexed local lives in scratch memory — 200 B per lane, 59 MB of scratch stores per launch — and
    // the launch took 29 us in the step's trace.)
    __shared__ float lds[kPackLdsFloats];
    constexpr int per_model = pack_blocks(PREC, true, true);
    const int bb = (int)blockIdx.x - draw_blocks;
    const int m = bb / per_model, b = bb - m * per_model;
    if (m >= n_models) return;
    pack_model_block<PREC>(P.P[m], P.packed[m], P.packed_bwd[m], b, lds);
}

}  // namespace nerfhip

extern "C" int nerfhip_train_prologue(const nerfhip_draw* draws_host, int n_draws, const nerfhip_ray_batch* batch_host, uint64_t seed,
                                      uint64_t offset, uint64_t* state, int max_blocks, uint64_t* increment_host,
                                      const float* const* weights_host, const float* const* biases_host, void* const* packed_host,
                                      void* const* packed_bwd_host, int n_models, int dtype, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(weights_host && biases_host && packed_host && packed_bwd_host);
    NERFHIP_CHECK_ARG(n_models >= 1 && n_models <= nerfhip::kPackMaxModels);
    if (dtype == NERFHIP_BF16_F8) dtype = NERFHIP_BF16;          // (the 8-bit mode shares the bf16 weight images)
    if (dtype != NERFHIP_F32 && dtype != NERFHIP_BF16) return NERFHIP_E_UNSUPPORTED;
    nerfhip::DrawTable T;
    int draw_blocks = 0;
    uint64_t inc = 0;
    const int rc = nerfhip_build_draw_table(draws_host, n_draws, batch_host, offset, state, max_blocks, &T, &draw_blocks, &inc);
    if (rc) return rc;
    if (increment_host) *increment_host = inc;
    nerfhip::MultiPackTable P;
    for (int m = 0; m < nerfhip::kPackMaxModels; ++m) {
        const int mm = m < n_models ? m : 0;
        NERFHIP_CHECK_ARG(packed_host[mm] && packed_bwd_host[mm]);
        if ((((uintptr_t)packed_host[mm]) | ((uintptr_t)packed_bwd_host[mm])) & 15) return NERFHIP_E_ALIGN;
        P.packed[m] = (uint8_t*)packed_host[mm];
        P.packed_bwd[m] = (uint8_t*)packed_bwd_host[mm];
        for (int i = 0; i < 12; ++i) {
            NERFHIP_CHECK_ARG(weights_host[12 * mm + i] && biases_host[12 * mm + i]);
            P.P[m].w[i] = weights_host[12 * mm + i];
            P.P[m].b[i] = biases_host[12 * mm + i];
        }
    }
    const dim3 grid((unsigned)(draw_blocks + n_models * nerfhip::pack_blocks(dtype, true, true)));
    if (dtype == NERFHIP_BF16)
        hipLaunchKernelGGL(nerfhip::train_prologue_kernel<NERFHIP_BF16>, grid, dim3(256), 0, (hipStream_t)stream, T, (unsigned long long)seed,
                           (unsigned long long)offset, reinterpret_cast<unsigned long long*>(state), draw_blocks, P, n_models);
    else
        hipLaunchKernelGGL(nerfhip::train_prologue_kernel<NERFHIP_F32>, grid, dim3(256), 0, (hipStream_t)stream, T, (unsigned long long)seed,
                           (unsigned long long)offset, reinterpret_cast<unsigned long long*>(state), draw_blocks, P, n_models);
    return nerfhip_launch_status();
}
