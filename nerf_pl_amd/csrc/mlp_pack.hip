// Repack the 24 nn.Linear tensors of a NeRF (reference models/nerf.py:60-81, state_dict order) into
// the MFMA A-fragment stream consumed by mlp_fwd (layout: mlp_layout.h).  Runs once per weight
// update; 0.6 M elements, negligible.
#include "common.h"
#include "mlp_layout.h"

namespace nerfhip {

struct ParamTable {
    const float* w[12];
    const float* b[12];
};

template <int PREC>
__global__ __launch_bounds__(64) void mlp_pack_kernel(ParamTable P, uint8_t* __restrict__ packed) {
    using namespace mlp;
    const int g = blockIdx.x;             // piece index
    const int lane = threadIdx.x;
    const int m = lane & 31, h = lane >> 5;
    uint4 outv = make_uint4(0, 0, 0, 0);
    if (g < total_pieces(PREC)) {
        int L = 0, start = 0;
        while (L + 1 < kNumLayers && g >= start + layer_pieces(L, PREC)) { start += layer_pieces(L, PREC); ++L; }
        const Layer ly = kLayers[L];
        const int rel = g - start;
        if (rel == 0) {
            // bias piece: 256 fp32, bias[m] for the layer's real outputs, zero padded
            const float* b = P.b[ly.param];
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane * 4 + q;
                v[q] = (idx < ly.n_out) ? b[idx] : 0.0f;
            }
            outv = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
        } else {
            const int f = (rel - 1) / ppf(PREC), sub = (rel - 1) % ppf(PREC);
            const int nks = ly.enc_slabs + ly.chain_slabs;
            const int t = frag_tile(f, ly.nt, nks), ks = frag_slab(f, ly.nt, nks);   // fragment order of mlp_fwd.hip run_layer
            const int row = 32 * t + m;
            const float* W = P.w[ly.param];
            const int ldw = kParamIn[ly.param];
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = layer_in_col(L, ks, h, j);
                v[j] = (row < ly.n_out && col >= 0) ? W[(size_t)row * ldw + col] : 0.0f;
            }
            if (PREC == NERFHIP_BF16) {
                typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
                bf16x8 p;
#pragma unroll
                for (int j = 0; j < 8; ++j) p[j] = (__bf16)v[j];     // round-to-nearest-even
                outv = *reinterpret_cast<uint4*>(&p);
            } else {
                outv = make_uint4(__float_as_uint(v[4 * sub + 0]), __float_as_uint(v[4 * sub + 1]),
                                  __float_as_uint(v[4 * sub + 2]), __float_as_uint(v[4 * sub + 3]));
            }
        }
    }
    reinterpret_cast<uint4*>(packed + (size_t)g * kPieceBytes)[lane] = outv;
}

}  // namespace nerfhip

extern "C" size_t nerfhip_mlp_packed_bytes(int dtype) {
    if (dtype == NERFHIP_BF16_F8) dtype = NERFHIP_BF16;          // same weight image: only the saved tensors differ
    if (dtype != NERFHIP_F32 && dtype != NERFHIP_BF16) return 0;
    return (size_t)nerfhip::mlp::padded_pieces(dtype) * nerfhip::mlp::kPieceBytes;
}

extern "C" int nerfhip_mlp_pack_weights(const float* const* weights_host, const float* const* biases_host,
                                        void* packed, int dtype, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(weights_host && biases_host && packed);
    if (dtype == NERFHIP_BF16_F8) dtype = NERFHIP_BF16;
    if (dtype != NERFHIP_F32 && dtype != NERFHIP_BF16) return NERFHIP_E_UNSUPPORTED;
    if (((uintptr_t)packed) & 15) return NERFHIP_E_ALIGN;
    nerfhip::ParamTable P;
    for (int i = 0; i < 12; ++i) {
        NERFHIP_CHECK_ARG(weights_host[i] && biases_host[i]);
        P.w[i] = weights_host[i];
        P.b[i] = biases_host[i];
    }
    const int n = nerfhip::mlp::padded_pieces(dtype);
    if (dtype == NERFHIP_BF16)
        hipLaunchKernelGGL(nerfhip::mlp_pack_kernel<NERFHIP_BF16>, dim3(n), dim3(64), 0, (hipStream_t)stream, P,
                           (uint8_t*)packed);
    else
        hipLaunchKernelGGL(nerfhip::mlp_pack_kernel<NERFHIP_F32>, dim3(n), dim3(64), 0, (hipStream_t)stream, P,
                           (uint8_t*)packed);
    return nerfhip_launch_status();
}
