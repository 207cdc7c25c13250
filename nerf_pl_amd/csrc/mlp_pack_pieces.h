// One 1 KiB piece of the packed weight streams (layout: mlp_layout.h), shared by the pack kernels of mlp_pack.hip:
//   pack_fwd_piece: the forward's A-fragment stream (bias piece + (tile, slab) fragments per layer, execution order)
//   pack_bwd_piece: the backward chain's W^T stream, followed by the fp32 fold block (mlp_layout.h kFold*: W_f, W_dx, b_f as
//                   mlp_bwd_fold_kernel reads them)
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace nerfhip {

struct ParamTable {
    const float* w[12];
    const float* b[12];
};
// several models per launch (a training step's coarse and fine network)
constexpr int kPackMaxModels = 4;
struct MultiPackTable {
    ParamTable P[kPackMaxModels];
    uint8_t* packed[kPackMaxModels];
    uint8_t* packed_bwd[kPackMaxModels];
};

template <int PREC>
__device__ __forceinline__ uint4 pack_fwd_piece(const ParamTable& P, int g, int lane) {
    using namespace mlp;
    const int m = lane & 31, h = lane >> 5;
    uint4 outv = make_uint4(0, 0, 0, 0);
    const int bb = bias_block_start(PREC);
    if (g >= bb && g < bb + bias_block_pieces(PREC)) {
        // bias block: piece L = the 256 fp32 of layer L's bias (real outputs, zero padded); the rest of the block is padding
        const int L = g - bb;
        if (L < kNumLayers) {
            const Layer ly = kLayers[L];
            const float* b = P.b[ly.param];
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane * 4 + q;
                v[q] = (idx < ly.n_out) ? b[idx] : 0.0f;
            }
            outv = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
        }
    } else if (g < total_pieces(PREC)) {
        int L = 0;
        while (L + 1 < kNumLayers && g >= layer_start(L + 1, PREC)) ++L;
        const Layer ly = kLayers[L];
        const int rel = g - layer_start(L, PREC);
        {
            const int f = rel / ppf(PREC), sub = rel % ppf(PREC);
            const int nks = ly.enc_slabs + ly.chain_slabs;
            const int t = frag_tile(f, ly.nt, nks), ks = frag_slab(f, ly.nt, nks);   // fragment order of mlp_fwd_kernel.h
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
    return outv;
}

template <int PREC>
__device__ __forceinline__ uint4 pack_bwd_piece(const ParamTable& P, int g, int lane) {
    using namespace mlp;
    const int m = lane & 31, h = lane >> 5;
    uint4 outv = make_uint4(0, 0, 0, 0);
    if (g >= bwd_padded_pieces(PREC)) {                             // fold block: plain fp32 rows, lane l = floats 4l .. 4l+3
        const int q = g - bwd_padded_pieces(PREC);
        const float* src = q < kFoldWdx ? P.w[8] + (size_t)(q - kFoldWf) * kParamIn[8]
                                        : (q < kFoldBf ? P.w[9] + (size_t)(q - kFoldWdx) * kParamIn[9] : P.b[8]);
        if (q < kFoldPieces)           // (W_dir rows are 283 floats: not 16-byte aligned, scalar loads)
            outv = make_uint4(__float_as_uint(src[4 * lane]), __float_as_uint(src[4 * lane + 1]), __float_as_uint(src[4 * lane + 2]),
                              __float_as_uint(src[4 * lane + 3]));
    } else if (g < bwd_total_pieces(PREC)) {
        int L = 0, start = 0;
        while (L + 1 < kNumBwdLayers && g >= start + bwd_layer_pieces(L, PREC)) { start += bwd_layer_pieces(L, PREC); ++L; }
        const BwdLayer ly = kBwdLayers[L];
        const int rel = g - start;
        const int f = rel / ppf(PREC), sub = rel % ppf(PREC);
        const int ks = bwd_frag_slab(f, ly.nt, ly.nks), t = bwd_frag_tile(f, ly.nt, ly.nks);
        const int icol = ly.col0 + 32 * t + m;                     // input feature of W == output row of W^T
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = 0.0f;
            if (ly.sigma_slab && ks == ly.nks - 1) {                // sigma head: one real row (W_sigma[0][:])
                if (h == 0 && j == 0) v[j] = P.w[10][icol];
            } else {
                const int o = chain_feature(ks, h, j);
                if (o < kParamOut[ly.param] && icol < kParamIn[ly.param])
                    v[j] = P.w[ly.param][(size_t)o * kParamIn[ly.param] + icol];
            }
        }
        if (PREC == NERFHIP_BF16) {
            typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
            bf16x8 p;
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = (__bf16)v[j];
            outv = *reinterpret_cast<uint4*>(&p);
        } else {
            outv = make_uint4(__float_as_uint(v[4 * sub + 0]), __float_as_uint(v[4 * sub + 1]),
                              __float_as_uint(v[4 * sub + 2]), __float_as_uint(v[4 * sub + 3]));
        }
    }
    return outv;
}

}  // namespace nerfhip
