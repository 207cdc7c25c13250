// K0 / K3: ray depth sampling.
//   sample_coarse_z   : rendering.py:183-204  (linspace / disparity + stratified jitter)
//   searchsorted_right: torchsearchsorted.searchsorted(side='right'), rendering.py:2,42 — the one
//                       native extension of the reference, re-done for gfx950
//   sample_pdf        : rendering.py:14-55   (pdf -> cdf -> inverse-CDF lerp)
//   fine_z            : rendering.py:223-229 (z_mid + sample_pdf + sort(cat)) in ONE launch
// All of these are HBM/latency-bound integer+float work: one ray per 64-lane wavefront, the ray's
// cdf/bins staged in LDS, coalesced row loads, no re-reads.  Scans run in fp64 and round per
// element, which is what torch-CPU cumsum does on fp32 data, so that the searchsorted
// indices track the CPU oracle.
#include "sampling_wave.h"

namespace nerfhip {

__global__ __launch_bounds__(256) void sample_coarse_z_kernel(const float* __restrict__ rays,
                                                               const float* __restrict__ prand,
                                                               float* __restrict__ z, int64_t B, int S, int use_disp,
                                                               float perturb) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * S) return;
    const int64_t r = idx / S;
    const int i = (int)(idx - r * S);
    const float near = rays[r * 8 + 6], far = rays[r * 8 + 7];
    const float zi = coarse_z_sample(near, far, i, S, use_disp, perturb, perturb > 0.0f ? prand[idx] : 0.0f);
    z[idx] = zi;
}

template <bool RIGHT>
__global__ __launch_bounds__(256) void searchsorted_kernel(const float* __restrict__ a,
                                                            const float* __restrict__ v,
                                                            int64_t* __restrict__ idx, int64_t B, int M, int K) {
    // one wave per row: the row of `a` is staged in LDS once, the K queries stream through coalesced.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= B) return;
    float* row = lds + (size_t)wave * M;
    for (int j = lane; j < M; j += 64) row[j] = a[r * M + j];
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < K; k += 64) {
        const float q = v[r * K + k];
        idx[r * K + k] = (int64_t)(RIGHT ? upper_bound(row, M, q) : lower_bound(row, M, q));
    }
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ bins, int64_t bins_stride,
                                                          const float* __restrict__ weights, int64_t w_stride,
                                                          const float* __restrict__ u, int64_t u_stride,
                                                          float* __restrict__ samples, int64_t B, int M, int K,
                                                          float eps, float* __restrict__ cdf_out,
                                                          int64_t* __restrict__ inds_out, int row_total) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= B) return;
    float* cdf_s = lds + (size_t)wave * 2 * (M + 1);
    float* bins_s = cdf_s + (M + 1);
    for (int j = lane; j <= M; j += 64) bins_s[j] = bins[r * bins_stride + j];
    const float* wrow = weights + r * w_stride;
    build_cdf_wave([&](int j) { return wrow[j]; }, M, eps, cdf_s, lane, row_total);
    if (cdf_out)
        for (int j = lane; j <= M; j += 64) cdf_out[r * (M + 1) + j] = cdf_s[j];
    for (int k = lane; k < K; k += 64) {
        const float uk = u ? u[r * u_stride + k] : linspace01(k, K);
        int ind;
        samples[r * K + k] = invert_cdf(cdf_s, bins_s, M, uk, eps, &ind);
        if (inds_out) inds_out[r * K + k] = (int64_t)ind;
    }
}

__global__ __launch_bounds__(256) void fine_z_kernel(const float* __restrict__ zc, const float* __restrict__ wc,
                                                      const float* __restrict__ u, int64_t u_stride,
                                                      float* __restrict__ zf, float* __restrict__ znew_out,
                                                      int64_t B, int S, int N, float eps, float* __restrict__ cdf_out,
                                                      int64_t* __restrict__ inds_out, int row_total) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= B) return;
    const float* wrow = wc + r * S + 1;         // weights_coarse[:, 1:-1]          :225
    fine_z_wave(lds + (size_t)wave * fine_z_lds_floats(S, N), zc + r * S, [&](int j) { return wrow[j]; },
                u ? u + r * u_stride : nullptr, S, N, eps, zf + r * (S + N), znew_out ? znew_out + r * N : nullptr,
                cdf_out ? cdf_out + r * (S - 1) : nullptr, inds_out ? inds_out + r * N : nullptr, lane, row_total);
}

}  // namespace nerfhip

extern "C" int nerfhip_sample_coarse_z(const float* rays, const float* perturb_rand, float* z, int64_t B, int S,
                                       int use_disp, float perturb, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S >= 1);
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(rays && z && (perturb <= 0.0f || perturb_rand));
    const int64_t total = B * S;
    hipLaunchKernelGGL(nerfhip::sample_coarse_z_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, rays, perturb_rand, z, B, S, use_disp, perturb);
    return nerfhip_launch_status();
}

static int searchsorted_impl(const float* a, const float* v, int64_t* idx, int64_t B, int M, int K, bool right,
                             nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && M >= 0 && K >= 0 && M <= 4096);
    if (B == 0 || K == 0) return 0;
    NERFHIP_CHECK_ARG(v && idx && (M == 0 || a));
    dim3 grid((unsigned)((B + 3) / 4)), block(256);
    const size_t lds = (size_t)4 * M * sizeof(float);
    if (right)
        hipLaunchKernelGGL(nerfhip::searchsorted_kernel<true>, grid, block, lds, (hipStream_t)stream, a, v, idx, B, M, K);
    else
        hipLaunchKernelGGL(nerfhip::searchsorted_kernel<false>, grid, block, lds, (hipStream_t)stream, a, v, idx, B, M, K);
    return nerfhip_launch_status();
}
extern "C" int nerfhip_searchsorted_right(const float* a, const float* v, int64_t* idx, int64_t B, int M, int K,
                                          nerfhip_stream_t stream) {
    return searchsorted_impl(a, v, idx, B, M, K, true, stream);
}
extern "C" int nerfhip_searchsorted_left(const float* a, const float* v, int64_t* idx, int64_t B, int M, int K,
                                         nerfhip_stream_t stream) {
    return searchsorted_impl(a, v, idx, B, M, K, false, stream);
}

extern "C" int nerfhip_sample_pdf_ex(const float* bins, int64_t bins_stride, const float* weights, int64_t w_stride,
                                     const float* u, int64_t u_stride, float* samples, int64_t B, int M, int K,
                                     float eps, float* cdf_out, int64_t* inds_out, int row_total, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && M >= 1 && K >= 0 && M <= 2040 && (row_total == 0 || row_total == 1));
    if (B == 0 || K == 0) return 0;
    NERFHIP_CHECK_ARG(bins && weights && samples);
    hipLaunchKernelGGL(nerfhip::sample_pdf_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256),
                       (size_t)4 * 2 * (M + 1) * sizeof(float), (hipStream_t)stream, bins, bins_stride, weights,
                       w_stride, u, u_stride, samples, B, M, K, eps, cdf_out, inds_out, row_total);
    return nerfhip_launch_status();
}
extern "C" int nerfhip_sample_pdf(const float* bins, int64_t bins_stride, const float* weights, int64_t w_stride,
                                  const float* u, int64_t u_stride, float* samples, int64_t B, int M, int K,
                                  float eps, nerfhip_stream_t stream) {
    return nerfhip_sample_pdf_ex(bins, bins_stride, weights, w_stride, u, u_stride, samples, B, M, K, eps, nullptr, nullptr,
                                 NERFHIP_ROW_TOTAL_EXACT, stream);
}

extern "C" int nerfhip_fine_z_ex(const float* z_coarse, const float* w_coarse, const float* u, int64_t u_stride,
                                 float* z_fine, float* z_new, int64_t B, int S_c, int N_i, float eps, float* cdf_out,
                                 int64_t* inds_out, int row_total, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S_c >= 3 && N_i >= 1 && (row_total == 0 || row_total == 1));
    const int S4 = (S_c + 3) & ~3, N4 = (N_i + 3) & ~3;
    const size_t per_wave = (size_t)(3 * S4 + N4 + ((S_c + 1 + 3) & ~3) + N4) * sizeof(float);
    NERFHIP_CHECK_ARG(4 * per_wave <= 65536);
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(z_coarse && w_coarse && z_fine);
    hipLaunchKernelGGL(nerfhip::fine_z_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 4 * per_wave,
                       (hipStream_t)stream, z_coarse, w_coarse, u, u_stride, z_fine, z_new, B, S_c, N_i, eps, cdf_out, inds_out, row_total);
    return nerfhip_launch_status();
}
extern "C" int nerfhip_fine_z(const float* z_coarse, const float* w_coarse, const float* u, int64_t u_stride,
                              float* z_fine, float* z_new, int64_t B, int S_c, int N_i, float eps,
                              nerfhip_stream_t stream) {
    return nerfhip_fine_z_ex(z_coarse, w_coarse, u, u_stride, z_fine, z_new, B, S_c, N_i, eps, nullptr, nullptr, NERFHIP_ROW_TOTAL_EXACT, stream);
}
