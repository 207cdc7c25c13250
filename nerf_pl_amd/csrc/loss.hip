// N2: the step after the path — MSELoss.forward (reference losses.py:9-14), psnr (metrics.py:4-13)
// and the seed of the backward pass in ONE launch instead of ~14 tiny ATen launches (two mse, two means, add,
// pow/log10/mul for the PSNR, and mse_backward x2 + fills in backward):
//     loss  = mean((rgb_coarse - t)^2) [+ mean((rgb_fine - t)^2)]
//     psnr  = -10 log10(mean((rgb_fine|coarse - t)^2))
//     g_c   = 2 (rgb_coarse - t) / n,   g_f = 2 (rgb_fine - t) / n          (d loss / d rgb, n = numel)
// n = 3 * rays is tiny (3072 floats at 1024 rays): one 1024-thread workgroup, grid-stride loop, LDS tree
// reduction in a fixed order (deterministic).  Latency-bound; the point is the launch count.
#include "common.h"

namespace nerfhip {

__global__ __launch_bounds__(1024) void mse_psnr_kernel(const float* __restrict__ rgb_c, const float* __restrict__ rgb_f,
                                                        const float* __restrict__ target, int64_t n,
                                                        float* __restrict__ out3, float* __restrict__ g_c,
                                                        float* __restrict__ g_f) {
    __shared__ float red[2][16];
    const int tid = threadIdx.x;
    const float scale = 2.0f / (float)n;
    float sc = 0.f, sf = 0.f;
    for (int64_t i = tid; i < n; i += 1024) {
        const float t = target[i];
        const float dc = nh_sub(rgb_c[i], t);
        sc += nh_mul(dc, dc);
        if (g_c) g_c[i] = nh_mul(dc, scale);
        if (rgb_f) {
            const float df = nh_sub(rgb_f[i], t);
            sf += nh_mul(df, df);
            if (g_f) g_f[i] = nh_mul(df, scale);
        }
    }
    sc = wave_sum(sc);
    sf = wave_sum(sf);
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = sc;
        red[1][tid >> 6] = sf;
    }
    __syncthreads();
    if (tid == 0) {
        float tc = 0.f, tf = 0.f;
        for (int w = 0; w < 16; ++w) {
            tc += red[0][w];
            tf += red[1][w];
        }
        const float mc = tc / (float)n, mf = tf / (float)n;
        out3[0] = rgb_f ? mc + mf : mc;                       // losses.py:10-13
        out3[1] = -10.0f * log10f(rgb_f ? mf : mc);           // metrics.py:12-13 on the fine (else coarse) image
        out3[2] = rgb_f ? mf : mc;
    }
}

}  // namespace nerfhip

extern "C" int nerfhip_mse_psnr(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n,
                                float* out3, float* g_coarse, float* g_fine, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n > 0 && rgb_coarse && target && out3);
    hipLaunchKernelGGL(nerfhip::mse_psnr_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rgb_coarse, rgb_fine, target, n,
                       out3, g_coarse, g_fine);
    return nerfhip_launch_status();
}
