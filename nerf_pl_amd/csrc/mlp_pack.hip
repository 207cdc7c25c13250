// Repack the 24 nn.Linear tensors of a NeRF (reference models/nerf.py:60-81, state_dict order) into the MFMA A-fragment
// streams consumed by the fused MLP kernels (layout: mlp_layout.h; one piece per 64-lane workgroup: mlp_pack_pieces.h):
// the forward stream, the backward chain's W^T stream, or — for a training step — both in ONE launch.  Runs once per weight
// update; 0.6 M elements, a few microseconds.
#include "mlp_pack_pieces.h"

namespace nerfhip {

// grids: pack_blocks() workgroups of 256 threads per model (mlp_pack_pieces.h pack_model_block: the W_c tiles, then four pieces each)
template <int PREC>
__global__ __launch_bounds__(kPackThreads) void mlp_pack_kernel(ParamTable P, uint8_t* __restrict__ packed) {
    __shared__ float lds[kPackLdsFloats];
    pack_model_block<PREC>(P, packed, nullptr, blockIdx.x, lds);
}
template <int PREC>
__global__ __launch_bounds__(kPackThreads) void mlp_pack_bwd_kernel(ParamTable P, uint8_t* __restrict__ packed) {
    __shared__ float lds[kPackLdsFloats];
    pack_model_block<PREC>(P, nullptr, packed, blockIdx.x, lds);
}
template <int PREC>
__global__ __launch_bounds__(kPackThreads) void mlp_pack_train_kernel(ParamTable P, uint8_t* __restrict__ packed,
                                                                      uint8_t* __restrict__ packed_bwd) {
    __shared__ float lds[kPackLdsFloats];
    pack_model_block<PREC>(P, packed, packed_bwd, blockIdx.x, lds);
}

// the same for up to kPackMaxModels models in ONE launch (blockIdx.y = model): a training step's coarse and fine network
template <int PREC>
__global__ __launch_bounds__(kPackThreads) void mlp_pack_train_multi_kernel(MultiPackTable T) {
    __shared__ float lds[kPackLdsFloats];
    const int m = blockIdx.y;
    pack_model_block<PREC>(T.P[m], T.packed[m], T.packed_bwd[m], blockIdx.x, lds);
}

static int pack_prec(int dtype) {       // NERFHIP_BF16_F8 shares the bf16 weight images: only the saved tensors differ
    if (dtype == NERFHIP_BF16_F8) return NERFHIP_BF16;
    return (dtype == NERFHIP_F32 || dtype == NERFHIP_BF16) ? dtype : -1;
}

}  // namespace nerfhip

extern "C" size_t nerfhip_mlp_packed_bytes(int dtype) {
    dtype = nerfhip::pack_prec(dtype);
    if (dtype < 0) return 0;
    return (size_t)nerfhip::mlp::padded_pieces(dtype) * nerfhip::mlp::kPieceBytes;
}
extern "C" size_t nerfhip_mlp_packed_bwd_bytes(int dtype) {
    dtype = nerfhip::pack_prec(dtype);
    if (dtype < 0) return 0;
    return (size_t)nerfhip::mlp::bwd_image_pieces(dtype) * nerfhip::mlp::kPieceBytes;     // W^T stream + fp32 fold block
}

extern "C" int nerfhip_mlp_pack_weights(const float* const* weights_host, const float* const* biases_host,
                                        void* packed, int dtype, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(weights_host && biases_host && packed);
    dtype = nerfhip::pack_prec(dtype);
    if (dtype < 0) return NERFHIP_E_UNSUPPORTED;
    if (((uintptr_t)packed) & 15) return NERFHIP_E_ALIGN;
    nerfhip::ParamTable P;
    for (int i = 0; i < 12; ++i) {
        NERFHIP_CHECK_ARG(weights_host[i] && biases_host[i]);
        P.w[i] = weights_host[i];
        P.b[i] = biases_host[i];
    }
    const int n = nerfhip::pack_blocks(dtype, true, false);
    if (dtype == NERFHIP_BF16)
        hipLaunchKernelGGL(nerfhip::mlp_pack_kernel<NERFHIP_BF16>, dim3(n), dim3(nerfhip::kPackThreads), 0, (hipStream_t)stream, P,
                           (uint8_t*)packed);
    else
        hipLaunchKernelGGL(nerfhip::mlp_pack_kernel<NERFHIP_F32>, dim3(n), dim3(nerfhip::kPackThreads), 0, (hipStream_t)stream, P,
                           (uint8_t*)packed);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_mlp_pack_weights_bwd(const float* const* weights_host, const float* const* biases_host, void* packed_bwd,
                                            int dtype, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(weights_host && biases_host && packed_bwd);
    dtype = nerfhip::pack_prec(dtype);
    if (dtype < 0) return NERFHIP_E_UNSUPPORTED;
    if (((uintptr_t)packed_bwd) & 15) return NERFHIP_E_ALIGN;
    nerfhip::ParamTable P;
    for (int i = 0; i < 12; ++i) {
        NERFHIP_CHECK_ARG(weights_host[i] && biases_host[i]);
        P.w[i] = weights_host[i];
        P.b[i] = biases_host[i];
    }
    const int n = nerfhip::pack_blocks(dtype, false, true);
    if (dtype == NERFHIP_BF16)
        hipLaunchKernelGGL(nerfhip::mlp_pack_bwd_kernel<NERFHIP_BF16>, dim3(n), dim3(nerfhip::kPackThreads), 0, (hipStream_t)stream, P, (uint8_t*)packed_bwd);
    else
        hipLaunchKernelGGL(nerfhip::mlp_pack_bwd_kernel<NERFHIP_F32>, dim3(n), dim3(nerfhip::kPackThreads), 0, (hipStream_t)stream, P, (uint8_t*)packed_bwd);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_mlp_pack_weights_train(const float* const* weights_host, const float* const* biases_host, void* packed,
                                              void* packed_bwd, int dtype, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(weights_host && biases_host && packed && packed_bwd);
    dtype = nerfhip::pack_prec(dtype);
    if (dtype < 0) return NERFHIP_E_UNSUPPORTED;
    if ((((uintptr_t)packed) | ((uintptr_t)packed_bwd)) & 15) return NERFHIP_E_ALIGN;
    nerfhip::ParamTable P;
    for (int i = 0; i < 12; ++i) {
        NERFHIP_CHECK_ARG(weights_host[i] && biases_host[i]);
        P.w[i] = weights_host[i];
        P.b[i] = biases_host[i];
    }
    const int n = nerfhip::pack_blocks(dtype, true, true);
    if (dtype == NERFHIP_BF16)
        hipLaunchKernelGGL(nerfhip::mlp_pack_train_kernel<NERFHIP_BF16>, dim3(n), dim3(nerfhip::kPackThreads), 0, (hipStream_t)stream, P,
                           (uint8_t*)packed, (uint8_t*)packed_bwd);
    else
        hipLaunchKernelGGL(nerfhip::mlp_pack_train_kernel<NERFHIP_F32>, dim3(n), dim3(nerfhip::kPackThreads), 0, (hipStream_t)stream, P,
                           (uint8_t*)packed, (uint8_t*)packed_bwd);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_mlp_pack_weights_train_multi(const float* const* weights_host, const float* const* biases_host,
                                                    void* const* packed_host, void* const* packed_bwd_host, int n_models, int dtype,
                                                    nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(weights_host && biases_host && packed_host && packed_bwd_host);
    NERFHIP_CHECK_ARG(n_models >= 1 && n_models <= nerfhip::kPackMaxModels);
    dtype = nerfhip::pack_prec(dtype);
    if (dtype < 0) return NERFHIP_E_UNSUPPORTED;
    nerfhip::MultiPackTable T;
    for (int m = 0; m < nerfhip::kPackMaxModels; ++m) {
        const int mm = m < n_models ? m : 0;
        NERFHIP_CHECK_ARG(packed_host[mm] && packed_bwd_host[mm]);
        if ((((uintptr_t)packed_host[mm]) | ((uintptr_t)packed_bwd_host[mm])) & 15) return NERFHIP_E_ALIGN;
        T.packed[m] = (uint8_t*)packed_host[mm];
        T.packed_bwd[m] = (uint8_t*)packed_bwd_host[mm];
        for (int i = 0; i < 12; ++i) {
            NERFHIP_CHECK_ARG(weights_host[12 * mm + i] && biases_host[12 * mm + i]);
            T.P[m].w[i] = weights_host[12 * mm + i];
            T.P[m].b[i] = biases_host[12 * mm + i];
        }
    }
    const dim3 grid((unsigned)nerfhip::pack_blocks(dtype, true, true), (unsigned)n_models);
    if (dtype == NERFHIP_BF16)
        hipLaunchKernelGGL(nerfhip::mlp_pack_train_multi_kernel<NERFHIP_BF16>, grid, dim3(nerfhip::kPackThreads), 0, (hipStream_t)stream, T);
    else
        hipLaunchKernelGGL(nerfhip::mlp_pack_train_multi_kernel<NERFHIP_F32>, grid, dim3(nerfhip::kPackThreads), 0, (hipStream_t)stream, T);
    return nerfhip_launch_status();
}
