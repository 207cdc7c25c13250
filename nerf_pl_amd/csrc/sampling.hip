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
#include "common.h"

// hipcc defaults to -ffp-contract=fast for device code; the reference computes mul and add as separate
// fp32 roundings (eager ATen ops), so fusing them would break bit-tracking of z / xyz / alpha.
#pragma clang fp contract(off)

namespace nerfhip {

// torch.linspace(0,1,S)[i] in fp32, bit for bit (checked against ATen for S in 7..192): symmetric
// form — first half step*i, second half end - step*(S-1-i) evaluated with ONE rounding (ATen's
// vectorised kernel fuses it) — so [S-1] == 1.0f exactly.
__device__ __forceinline__ float linspace01(int i, int S) {
    if (S <= 1) return 0.0f;
    const float step = __fdiv_rn(1.0f, (float)(S - 1));
    if (i < S / 2) return __fmul_rn(step, (float)i);
    return __builtin_fmaf(-step, (float)(S - 1 - i), 1.0f);
}

__device__ __forceinline__ float coarse_z_raw(float near, float far, int i, int S, int use_disp) {
    const float t = linspace01(i, S);
    const float omt = __fsub_rn(1.0f, t);
    if (!use_disp) return __fadd_rn(__fmul_rn(near, omt), __fmul_rn(far, t));                  // :191
    const float a = __fmul_rn(__fdiv_rn(1.0f, near), omt), b = __fmul_rn(__fdiv_rn(1.0f, far), t);  // :193
    return __fdiv_rn(1.0f, __fadd_rn(a, b));
}

__global__ __launch_bounds__(256) void sample_coarse_z_kernel(const float* __restrict__ rays,
                                                               const float* __restrict__ prand,
                                                               float* __restrict__ z, int64_t B, int S, int use_disp,
                                                               float perturb) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * S) return;
    const int64_t r = idx / S;
    const int i = (int)(idx - r * S);
    const float near = rays[r * 8 + 6], far = rays[r * 8 + 7];
    float zi = coarse_z_raw(near, far, i, S, use_disp);
    if (perturb > 0.0f) {  // :197-204
        const float zl = (i > 0) ? coarse_z_raw(near, far, i - 1, S, use_disp) : zi;
        const float zr = (i < S - 1) ? coarse_z_raw(near, far, i + 1, S, use_disp) : zi;
        const float lower = (i > 0) ? __fmul_rn(0.5f, __fadd_rn(zl, zi)) : zi;
        const float upper = (i < S - 1) ? __fmul_rn(0.5f, __fadd_rn(zi, zr)) : zi;
        const float pr = __fmul_rn(perturb, prand[idx]);
        zi = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), pr));
    }
    z[idx] = zi;
}

// first j in [0,n] with a[j] > v   (numpy side='right')
// first j in [0,n] with a[j] >= v  (numpy side='left')
template <typename P>
__device__ __forceinline__ int lower_bound(P a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <typename P>
__device__ __forceinline__ int upper_bound(P a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
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

// ---- shared per-wave inverse-CDF machinery ---------------------------------------------------
// cdf_s: M+1 floats in LDS (built here), bins_s: M+1 floats in LDS (filled by the caller).
template <typename WLoad>
__device__ __forceinline__ void build_cdf_wave(WLoad wload, int M, float eps, float* cdf_s, int lane) {
    // weights + eps, total (fp64 sum of the fp32 terms, rounded once)            rendering.py:29-30
    double part = 0.0;
    for (int j = lane; j < M; j += 64) part += (double)__fadd_rn(wload(j), eps);
    const float total = (float)wave_sum(part);
    // cdf = [0, cumsum(pdf)]                                                      :31-33
    double carry = 0.0;
    for (int j0 = 0; j0 < M; j0 += 64) {
        const int j = j0 + lane;
        const float pdf = (j < M) ? __fdiv_rn(__fadd_rn(wload(j), eps), total) : 0.0f;
        const double incl = wave_incl_sum((double)pdf, lane) + carry;
        if (j < M) cdf_s[j + 1] = (float)incl;
        carry = __shfl(incl, 63, 64);
    }
    if (lane == 0) cdf_s[0] = 0.0f;
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float invert_cdf(const float* cdf_s, const float* bins_s, int M, float u, float eps) {
    const int ind = upper_bound(cdf_s, M + 1, u);   // searchsorted(cdf,u,'right')   :42
    const int below = max(ind - 1, 0);              // :43
    const int above = min(ind, M);                  // :44
    const float cb = cdf_s[below], ca = cdf_s[above];
    const float bb = bins_s[below], ba = bins_s[above];
    float denom = __fsub_rn(ca, cb);
    if (denom < eps) denom = 1.0f;                  // :51
    const float t = __fdiv_rn(__fsub_rn(u, cb), denom);
    return __fadd_rn(bb, __fmul_rn(t, __fsub_rn(ba, bb)));  // :54
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ bins, int64_t bins_stride,
                                                          const float* __restrict__ weights, int64_t w_stride,
                                                          const float* __restrict__ u, int64_t u_stride,
                                                          float* __restrict__ samples, int64_t B, int M, int K,
                                                          float eps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= B) return;
    float* cdf_s = lds + (size_t)wave * 2 * (M + 1);
    float* bins_s = cdf_s + (M + 1);
    for (int j = lane; j <= M; j += 64) bins_s[j] = bins[r * bins_stride + j];
    const float* wrow = weights + r * w_stride;
    build_cdf_wave([&](int j) { return wrow[j]; }, M, eps, cdf_s, lane);
    for (int k = lane; k < K; k += 64) {
        const float uk = u ? u[r * u_stride + k] : linspace01(k, K);
        samples[r * K + k] = invert_cdf(cdf_s, bins_s, M, uk, eps);
    }
}

// z_fine = sort(cat(z_coarse, sample_pdf(z_mid, w[:,1:-1], N_i)))        rendering.py:223-229
__global__ __launch_bounds__(256) void fine_z_kernel(const float* __restrict__ zc, const float* __restrict__ wc,
                                                      const float* __restrict__ u, int64_t u_stride,
                                                      float* __restrict__ zf, float* __restrict__ znew_out,
                                                      int64_t B, int S, int N, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= B) return;
    const int M = S - 2;                        // number of pdf bins
    float* zc_s = lds + (size_t)wave * (3 * S + N);
    float* cdf_s = zc_s + S;                    // M+1 = S-1
    float* bins_s = cdf_s + S;                  // M+1 = S-1 midpoints
    float* zn_s = bins_s + S;                   // N new samples
    for (int j = lane; j < S; j += 64) zc_s[j] = zc[r * S + j];
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < S - 1; j += 64) bins_s[j] = __fmul_rn(0.5f, __fadd_rn(zc_s[j], zc_s[j + 1]));  // :223
    const float* wrow = wc + r * S + 1;         // weights_coarse[:, 1:-1]          :225
    build_cdf_wave([&](int j) { return wrow[j]; }, M, eps, cdf_s, lane);
    for (int k = lane; k < N; k += 64) {
        const float uk = u ? u[r * u_stride + k] : linspace01(k, N);
        const float v = invert_cdf(cdf_s, bins_s, M, uk, eps);
        zn_s[k] = v;
        if (znew_out) znew_out[r * N + k] = v;
    }
    __builtin_amdgcn_wave_barrier();
    // Stable rank merge of the concatenation [zc (sorted) | zn (any order)].  NaNs are not ordered.
    float* out = zf + r * (S + N);
    for (int i = lane; i < S; i += 64) {        // coarse element i: earlier in cat order => wins ties
        const float x = zc_s[i];
        int less = 0;
        for (int k = 0; k < N; ++k) less += (zn_s[k] < x) ? 1 : 0;
        // coarse samples are non-decreasing, so #coarse placed before element i is i itself
        out[i + less] = x;
    }
    for (int k = lane; k < N; k += 64) {
        const float x = zn_s[k];
        int rank = upper_bound(zc_s, S, x);     // #coarse <= x
        for (int q = 0; q < N; ++q) {
            const float y = zn_s[q];
            rank += (y < x || (y == x && q < k)) ? 1 : 0;
        }
        out[rank] = x;
    }
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

extern "C" int nerfhip_sample_pdf(const float* bins, int64_t bins_stride, const float* weights, int64_t w_stride,
                                  const float* u, int64_t u_stride, float* samples, int64_t B, int M, int K,
                                  float eps, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && M >= 1 && K >= 0 && M <= 2040);
    if (B == 0 || K == 0) return 0;
    NERFHIP_CHECK_ARG(bins && weights && samples);
    hipLaunchKernelGGL(nerfhip::sample_pdf_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256),
                       (size_t)4 * 2 * (M + 1) * sizeof(float), (hipStream_t)stream, bins, bins_stride, weights,
                       w_stride, u, u_stride, samples, B, M, K, eps);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_fine_z(const float* z_coarse, const float* w_coarse, const float* u, int64_t u_stride,
                              float* z_fine, float* z_new, int64_t B, int S_c, int N_i, float eps,
                              nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S_c >= 3 && N_i >= 1 && (3 * S_c + N_i) <= 4096);
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(z_coarse && w_coarse && z_fine);
    hipLaunchKernelGGL(nerfhip::fine_z_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256),
                       (size_t)4 * (3 * S_c + N_i) * sizeof(float), (hipStream_t)stream, z_coarse, w_coarse, u,
                       u_stride, z_fine, z_new, B, S_c, N_i, eps);
    return nerfhip_launch_status();
}
