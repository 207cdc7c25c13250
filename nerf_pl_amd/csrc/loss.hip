// N2: the step after the path — MSELoss.forward (reference losses.py:9-14), psnr (metrics.py:4-13)
// and the seed of the backward pass in ONE launch instead of ~14 tiny ATen launches (two mse, two means, add,
// pow/log10/mul for the PSNR, and mse_backward x2 + fills in backward):
//     loss  = mean((rgb_coarse - t)^2) [+ mean((rgb_fine - t)^2)]
//     psnr  = -10 log10(mean((rgb_fine|coarse - t)^2))
//     g_c   = 2 (rgb_coarse - t) / n,   g_f = 2 (rgb_fine - t) / n          (d loss / d rgb, n = numel)
// n = 3 * rays is tiny (3072 floats at 1024 rays): one 1024-thread workgroup, grid-stride loop, LDS tree
// reduction in a fixed order (deterministic).  Latency-bound; the point is the launch count.
#include "loss_math.h"

namespace nerfhip {

__global__ __launch_bounds__(1024) void mse_psnr_kernel(const float* __restrict__ rgb_c, const float* __restrict__ rgb_f,
                                                        const float* __restrict__ target, int64_t n,
                                                        float* __restrict__ out3, float* __restrict__ g_c,
                                                        float* __restrict__ g_f) {
    __shared__ float red[2][16];
    mse_psnr_block<1>([&](int64_t i) { return rgb_c[i]; }, [&](int64_t i) { return rgb_f[i]; }, rgb_f != nullptr, target, n, out3, g_c, g_f, red);
}

}  // namespace nerfhip

extern "C" int nerfhip_mse_psnr(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n,
                                float* out3, float* g_coarse, float* g_fine, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n > 0 && rgb_coarse && target && out3);
    hipLaunchKernelGGL(nerfhip::mse_psnr_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rgb_coarse, rgb_fine, target, n,
                       out3, g_coarse, g_fine);
    return nerfhip_launch_status();
}
