// MSELoss.forward (reference losses.py:9-14) + psnr (metrics.py:4-13) + the seed of the backward pass as ONE workgroup's work,
// shared by mse_psnr_kernel (loss.hip, 1024 threads) and the compositing kernel that appends the loss to the fine pass's
// quadrature (composite.hip, the LAST of its 256-thread workgroups to finish):
//     loss  = mean((rgb_coarse - t)^2) [+ mean((rgb_fine - t)^2)]
//     psnr  = -10 log10(mean((rgb_fine|coarse - t)^2))
//     g_c   = 2 (rgb_coarse - t) / n,   g_f = 2 (rgb_fine - t) / n          (d loss / d rgb, n = numel)
// The reduction order is fixed and is that of 1024 threads: thread v sums elements v, v + 1024, ..., the 64 lanes of a wave
// are summed by a butterfly, the 16 wave totals sequentially.  A workgroup of 1024 / Q threads runs it with every thread
// standing in for Q of the 1024 (same lane, waves w, w + 16 / Q, ...): the same additions in the same order, the same bits.
#pragma once
#include "common.h"

namespace nerfhip {

template <int Q, typename LoadC, typename LoadF>
__device__ __forceinline__ void mse_psnr_block(LoadC load_c, LoadF load_f, bool have_f,
                                               const float* __restrict__ target, int64_t n, float* __restrict__ out3,
                                               float* __restrict__ g_c, float* __restrict__ g_f, float (*red)[16]) {
    constexpr int NT = 1024 / Q;
    constexpr int J = 4;              // elements per stand-in thread fetched up front (n <= 4096 floats: 1365 rays, the training batch)
    const int tid = threadIdx.x;
    const float scale = 2.0f / (float)n;
    // every load of the first J rounds is issued before the first addition: the loads of one thread are independent, the
    // additions are not (a 256-thread tail would otherwise pay Q * n / 1024 dependent memory round trips)
    float tv[Q][J], cv[Q][J], fv[Q][J];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int64_t i = tid + NT * q + 1024 * j;
            const bool ok = i < n;
            tv[q][j] = ok ? target[i] : 0.0f;
            cv[q][j] = ok ? load_c(i) : 0.0f;
            fv[q][j] = (ok && have_f) ? load_f(i) : 0.0f;
        }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int v = tid + NT * q;
        float sc = 0.f, sf = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int64_t i = v + 1024 * j;
            if (i < n) {
                const float dc = nh_sub(cv[q][j], tv[q][j]);
                sc += nh_mul(dc, dc);
                if (g_c) g_c[i] = nh_mul(dc, scale);
                if (have_f) {
                    const float df = nh_sub(fv[q][j], tv[q][j]);
                    sf += nh_mul(df, df);
                    if (g_f) g_f[i] = nh_mul(df, scale);
                }
            }
        }
        for (int64_t i = v + 1024 * J; i < n; i += 1024) {
            const float t = target[i];
            const float dc = nh_sub(load_c(i), t);
            sc += nh_mul(dc, dc);
            if (g_c) g_c[i] = nh_mul(dc, scale);
            if (have_f) {
                const float df = nh_sub(load_f(i), t);
                sf += nh_mul(df, df);
                if (g_f) g_f[i] = nh_mul(df, scale);
            }
        }
        sc = wave_sum(sc);
        sf = wave_sum(sf);
        if ((tid & 63) == 0) {
            red[0][v >> 6] = sc;
            red[1][v >> 6] = sf;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float tc = 0.f, tf = 0.f;
        for (int w = 0; w < 16; ++w) {
            tc += red[0][w];
            tf += red[1][w];
        }
        const float mc = tc / (float)n, mf = tf / (float)n;
        out3[0] = have_f ? mc + mf : mc;                       // losses.py:10-13
        out3[1] = -10.0f * log10f(have_f ? mf : mc);           // metrics.py:12-13 on the fine (else coarse) image
        out3[2] = have_f ? mf : mc;
    }
}

}  // namespace nerfhip
