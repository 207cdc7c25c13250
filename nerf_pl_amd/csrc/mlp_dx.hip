// d loss / d x for pre-embedded inputs of NeRF.forward (reference models/nerf.py:100-124 is an ordinary differentiable
// module: autograd gives dL/dx (n, 90) when x requires grad).  The reference's own training never asks for it — rays carry no
// gradient and the importance samples are detached (rendering.py:226) — so this is API completeness, not hot path: a plain
// FMA kernel over the dY slabs the backward chain already wrote,
//     dx[:, 0:63]  = W_1^T dY_1 + W_5[:, 0:63]^T dY_5        (xyz encoding enters layers 1 and 5, nerf.py:108-109)
//     dx[:, 63:90] = W_dir[:, 256:283]^T dY_dir               (direction encoding enters dir_encoding, nerf.py:118)
// one wave per 32-point tile, lane (n, h) holding the 128 features chain_feature(ks, h, j) of its point; the two halves are
// combined with one cross-lane add per channel.  Weight rows are broadcast loads (two distinct addresses per instruction).
#include <type_traits>

#include "common.h"
#include "mlp_layout.h"

namespace nerfhip {
using namespace mlp;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) float f32x8;

template <typename Slab> __device__ __forceinline__ float sget(const Slab& s, int j) { return (float)s[j]; }

template <int PREC>
__global__ __launch_bounds__(256) void mlp_dx_embedded_kernel(const uint8_t* __restrict__ dys, int64_t n, int64_t ntiles,
                                                              const float* __restrict__ w1, const float* __restrict__ w5,
                                                              const float* __restrict__ wdir, float* __restrict__ gx,
                                                              int64_t gx_stride) {
    using Slab = typename std::conditional<PREC == NERFHIP_BF16, bf16x8, f32x8>::type;
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int h = lane >> 5;
    const int64_t p = tile * 32 + (lane & 31);
    constexpr int IL = act_il(PREC);          // bf16 slabs: the block's pieces are IL KiB apart (mlp_layout.h)
    const uint8_t* tb = dys + tile_block_off(tile, kDySlabs * 64 * (int)sizeof(Slab), IL);
    auto slab = [&](int sec) { return *reinterpret_cast<const Slab*>(tb + ((size_t)sec * 64 * IL + lane) * sizeof(Slab)); };

    float ax[kXyzCh];
#pragma unroll
    for (int c = 0; c < kXyzCh; ++c) ax[c] = 0.0f;
    for (int ks = 0; ks < 16; ++ks) {
        const Slab g1 = slab(dy_h(1) + ks), g5 = slab(dy_h(5) + ks);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = chain_feature(ks, h, j);
            const float a = sget(g1, j), b = sget(g5, j);
            const float* r1 = w1 + (size_t)f * kParamIn[0];
            const float* r5 = w5 + (size_t)f * kParamIn[4];
#pragma unroll
            for (int c = 0; c < kXyzCh; ++c) ax[c] = __builtin_fmaf(a, r1[c], __builtin_fmaf(b, r5[c], ax[c]));
        }
    }
    float ad[kDirCh];
#pragma unroll
    for (int c = 0; c < kDirCh; ++c) ad[c] = 0.0f;
    for (int ks = 0; ks < 8; ++ks) {
        const Slab gd = slab(kDyDir + ks);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = chain_feature(ks, h, j);
            const float a = sget(gd, j);
            const float* rd = wdir + (size_t)f * kParamIn[9] + kW;
#pragma unroll
            for (int c = 0; c < kDirCh; ++c) ad[c] = __builtin_fmaf(a, rd[c], ad[c]);
        }
    }
    float* row = gx + (p < n ? p : 0) * gx_stride;
#pragma unroll
    for (int c = 0; c < kXyzCh; ++c) {
        const float v = ax[c] + __shfl_xor(ax[c], 32, 64);
        if (h == 0 && p < n) row[c] = v;
    }
#pragma unroll
    for (int c = 0; c < kDirCh; ++c) {
        const float v = ad[c] + __shfl_xor(ad[c], 32, 64);
        if (h == 0 && p < n) row[kXyzCh + c] = v;
    }
}

}  // namespace nerfhip

extern "C" int nerfhip_mlp_dx_embedded(const void* dys, int64_t n, const float* w_xyz1, const float* w_xyz5, const float* w_dir,
                                       float* gx, int64_t gx_stride, int dtype, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0 && gx_stride >= 90);
    if (dtype != NERFHIP_F32 && dtype != NERFHIP_BF16) return NERFHIP_E_UNSUPPORTED;     // e5m2 dY (NERFHIP_BF16_F8) is too coarse for dx
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(dys && w_xyz1 && w_xyz5 && w_dir && gx);
    const int64_t ppw = 32 * (dtype == NERFHIP_F32 ? 4 : 8);
    const int64_t tiles = (n + ppw - 1) / ppw * (ppw / 32);
    const unsigned blocks = (unsigned)((tiles + 3) / 4);
    if (dtype == NERFHIP_BF16)
        hipLaunchKernelGGL(nerfhip::mlp_dx_embedded_kernel<NERFHIP_BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint8_t*)dys, n, tiles, w_xyz1, w_xyz5, w_dir, gx, gx_stride);
    else
        hipLaunchKernelGGL(nerfhip::mlp_dx_embedded_kernel<NERFHIP_F32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint8_t*)dys, n, tiles, w_xyz1, w_xyz5, w_dir, gx, gx_stride);
    return nerfhip_launch_status();
}
