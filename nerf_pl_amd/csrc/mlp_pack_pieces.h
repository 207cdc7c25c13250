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

// ---- the folded layer's weights (mlp_layout.h kLayers): W_c = W_dir[:, :256] W_final (128 x 256), b_c = W_dir[:, :256] b_final + b_dir ----
// Formed by the pack kernels themselves, one 4-wave WORKGROUP per 32 x 32 tile of W_c on the fp32 MFMA (v_mfma_f32_32x32x2_f32: fmaf
// chains), and written from the tile into BOTH images — the forward's A fragments and the chain's W^T fragments take their bf16 (or
// fp32) values from the same fp32 numbers.  Operands in ONE memory round trip, every load independent and coalesced: the A tile
// W_dir[32 rt .., 0..255] row by row (256 B per instruction) into LDS, where the MFMA's A operand — lane (row r, inner index m + kh) —
// is a conflict-free read; the B operand W_final[m + kh][32 ct + r] straight into registers in the order the chain consumes it; each
// wave takes a quarter of the inner dimension, the partial tiles are summed in wave order.
// (History: per-element dot products inside pack_*_piece — 256 dependent load round trips per piece, 80 us in the training step's
// prologue launch; one wave per tile with per-step loads — 25-65 us per tile, the prologue at 22 us against 13.)
// The generic piece functions SKIP these pieces (fwd_piece_folded / bwd_piece_folded).
constexpr int kFoldTiles = 4 * 8;              // (dir-feature tile rt, h8-feature tile ct) = tile / 8, tile % 8
constexpr int kFoldPitch = 257;                // LDS pitch (floats) of the staged A tile
constexpr int kPackLdsFloats = 32 * kFoldPitch + 4 * 16 * 64 + 32 * 33 + 4 * 64;     // A tile | 4 partial C tiles | C tile | bias partials
constexpr int kPackThreads = 256;              // every pack kernel: 4 waves per workgroup = one W_c tile or four 1 KiB pieces
NH_HD constexpr bool fwd_piece_folded(int g, int prec) {
    using namespace mlp;
    const int g0 = layer_start(kDirLayer, prec), nks = layer_slabs(kDirLayer);
    if (g >= bias_block_start(prec) && g < bias_block_start(prec) + bias_block_pieces(prec)) return g - bias_block_start(prec) == kDirLayer;
    if (g < g0 || g >= g0 + layer_pieces(kDirLayer, prec)) return false;
    return ((g - g0) / ppf(prec)) % nks >= kLayers[kDirLayer].enc_slabs;
}
NH_HD constexpr bool bwd_piece_folded(int g, int prec) {
    using namespace mlp;
    const int g0 = bwd_layer_start(kBwdLayerFold, prec), nks = kBwdLayers[kBwdLayerFold].nks;
    if (g < g0 || g >= g0 + bwd_layer_pieces(kBwdLayerFold, prec)) return false;
    return ((g - g0) / ppf(prec)) % nks < nks - 1;             // (the last slab is the sigma head's row: generic)
}
// one 256-thread workgroup: tile `tile` of W_c -> its pieces of `packed` (forward image, may be null) and `packed_bwd` (W^T image,
// may be null); `lds`: kPackLdsFloats floats
template <int PREC>
__device__ __forceinline__ void pack_fold_tile(const ParamTable& P, uint8_t* __restrict__ packed, uint8_t* __restrict__ packed_bwd, int tile,
                                               float* lds) {
    using namespace mlp;
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int rt = tile >> 3, ct = tile & 7;
    const int r = lane & 31, kh = lane >> 5;
    float* sa = lds;                                       // [32][kFoldPitch]
    float* part = sa + 32 * kFoldPitch;                    // [4][16][64]
    float* sc = part + 4 * 16 * 64;                        // [32][33]
    float* pb = sc + 32 * 33;                              // [4][64]
    // A: this wave's 8 rows of the tile; B and b_final: this wave's quarter of the inner dimension, m = 64 wave + 2 j + kh
    const float* __restrict__ wd = P.w[9] + (size_t)(32 * rt + 8 * wave) * kParamIn[9] + lane;
    const float* __restrict__ wf = P.w[8] + (size_t)(64 * wave + kh) * kW + 32 * ct + r;
    const float* __restrict__ bfin = P.b[8] + 64 * wave + kh;
    float va[32], vb[32], vf[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) va[i] = wd[(size_t)(i >> 2) * kParamIn[9] + 64 * (i & 3)];
#pragma unroll
    for (int j = 0; j < 32; ++j) vb[j] = wf[(size_t)(2 * j) * kW];
#pragma unroll
    for (int j = 0; j < 32; ++j) vf[j] = bfin[2 * j];
#pragma unroll
    for (int i = 0; i < 32; ++i) sa[(8 * wave + (i >> 2)) * kFoldPitch + 64 * (i & 3) + lane] = va[i];
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    float bacc = 0.0f;
    const float* pa = sa + r * kFoldPitch + 64 * wave + kh;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float a = pa[2 * j];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vb[j], acc, 0, 0, 0);
        bacc = __builtin_fmaf(a, vf[j], bacc);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) part[(wave * 16 + q) * 64 + lane] = acc[q];
    pb[wave * 64 + lane] = bacc;
    __syncthreads();
    // C/D layout: lane -> column lane & 31, register q -> row (q & 3) + 8 (q >> 2) + 4 (lane >> 5); partials summed in wave order
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = wave + 4 * i;
        sc[((q & 3) + 8 * (q >> 2) + 4 * kh) * 33 + r] = ((part[q * 64 + lane] + part[(16 + q) * 64 + lane]) + part[(32 + q) * 64 + lane]) + part[(48 + q) * 64 + lane];
    }
    __syncthreads();
    const int h = kh;
    constexpr int PPF = ppf(PREC);
    auto emit = [&](uint8_t* img, int g0, const float (&v)[8]) {
        if (PREC == NERFHIP_BF16) {
            typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
            bf16x8 pk;
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[j] = (__bf16)v[j];
            reinterpret_cast<uint4*>(img + (size_t)g0 * kPieceBytes)[lane] = *reinterpret_cast<uint4*>(&pk);
        } else {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
                reinterpret_cast<uint4*>(img + (size_t)(g0 + sub) * kPieceBytes)[lane] =
                    make_uint4(__float_as_uint(v[4 * sub]), __float_as_uint(v[4 * sub + 1]), __float_as_uint(v[4 * sub + 2]), __float_as_uint(v[4 * sub + 3]));
        }
    };
    // the tile's four fragments — forward (slab s2 = 0, 1), W^T (s2 = 0, 1) — one per wave
    const int s2 = wave & 1;
    float v[8];
    if (wave < 2) {
        if (packed) {            // forward fragment (tile rt, slab enc + 2 ct + s2): lane (m, h) = row m, columns chain_feature(2 ct + s2, h, j)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = sc[r * 33 + 16 * s2 + 8 * (j >> 2) + 4 * h + (j & 3)];
            emit(packed, layer_start(kDirLayer, PREC) + (rt * layer_slabs(kDirLayer) + kLayers[kDirLayer].enc_slabs + 2 * ct + s2) * PPF, v);
        }
    } else if (packed_bwd) {     // W^T fragment (tile ct, slab 2 rt + s2): lane (m, h) = h8 feature m, dir features chain_feature(2 rt + s2, h, j)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sc[(16 * s2 + 8 * (j >> 2) + 4 * h + (j & 3)) * 33 + r];
        emit(packed_bwd, bwd_layer_start(kBwdLayerFold, PREC) + (ct * kBwdLayers[kBwdLayerFold].nks + 2 * rt + s2) * PPF, v);
    }
    if (packed && ct == 0 && wave == 0) {     // b_c rows 32 rt ..: the lane halves hold the even / odd m partial sums of each wave's quarter
        float* bias = reinterpret_cast<float*>(packed + (size_t)(bias_block_start(PREC) + kDirLayer) * kPieceBytes);
        if (kh == 0) {
            float sum = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) sum += pb[w * 64 + r] + pb[w * 64 + 32 + r];
            bias[32 * rt + r] = P.b[9][32 * rt + r] + sum;
        }
        if (rt == 0) bias[128 + 2 * lane] = bias[128 + 2 * lane + 1] = 0.0f;             // (outputs 128..255 of the piece: padding)
    }
}
// one 256-thread workgroup of a pack launch over ONE model: workgroups [0, kFoldTiles) = the W_c tiles (first: the longest units), then
// four 1 KiB pieces per workgroup over the forward image (if `packed`) and the W^T image (if `packed_bwd`)
NH_HD constexpr int pack_blocks(int prec, bool fwd, bool bwd) {
    return kFoldTiles + ((fwd ? mlp::padded_pieces(prec) : 0) + (bwd ? mlp::bwd_image_pieces(prec) : 0) + 3) / 4;
}
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
                v[j] = (row < ly.n_out && col >= 0) ? W[(size_t)row * ldw + col] : 0.0f;      // (folded pieces: never stored, see pack_fold_tile)
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
                if (o < kParamOut[ly.param] && icol < kParamIn[ly.param])       // (folded pieces: never stored, see pack_fold_tile)
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

template <int PREC>
__device__ __forceinline__ void pack_model_block(const ParamTable& P, uint8_t* __restrict__ packed, uint8_t* __restrict__ packed_bwd, int b, float* lds) {
    if (b < kFoldTiles) {
        pack_fold_tile<PREC>(P, packed, packed_bwd, b, lds);
        return;
    }
    const int nf = packed ? mlp::padded_pieces(PREC) : 0, nb = packed_bwd ? mlp::bwd_image_pieces(PREC) : 0;
    const int lane = threadIdx.x & 63;
    // the piece index is wave-uniform and SAID to be so (readfirstlane): the tables are then read with scalar loads
    const int g = __builtin_amdgcn_readfirstlane((b - kFoldTiles) * 4 + (int)(threadIdx.x >> 6));
    if (g < nf) {
        if (!fwd_piece_folded(g, PREC)) reinterpret_cast<uint4*>(packed + (size_t)g * mlp::kPieceBytes)[lane] = pack_fwd_piece<PREC>(P, g, lane);
    } else if (g < nf + nb) {
        if (!bwd_piece_folded(g - nf, PREC))
            reinterpret_cast<uint4*>(packed_bwd + (size_t)(g - nf) * mlp::kPieceBytes)[lane] = pack_bwd_piece<PREC>(P, g - nf, lane);
    }
}

}  // namespace nerfhip
