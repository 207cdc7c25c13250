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

namespace nerfhip {

// torch.linspace(0,1,S)[i] in fp32, bit for bit (checked against ATen for S in 7..192): symmetric
// form — first half step*i, second half end - step*(S-1-i) evaluated with ONE rounding (ATen's
// vectorised kernel fuses it) — so [S-1] == 1.0f exactly.
__device__ __forceinline__ float linspace01(int i, int S) {
    if (S <= 1) return 0.0f;
    const float step = nh_div(1.0f, (float)(S - 1));
    if (i < S / 2) return nh_mul(step, (float)i);
    return __builtin_fmaf(-step, (float)(S - 1 - i), 1.0f);
}

__device__ __forceinline__ float coarse_z_raw(float near, float far, int i, int S, int use_disp) {
    const float t = linspace01(i, S);
    const float omt = nh_sub(1.0f, t);
    if (!use_disp) return nh_add(nh_mul(near, omt), nh_mul(far, t));                  // :191
    const float a = nh_mul(nh_div(1.0f, near), omt), b = nh_mul(nh_div(1.0f, far), t);  // :193
    return nh_div(1.0f, nh_add(a, b));
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
        const float lower = (i > 0) ? nh_mul(0.5f, nh_add(zl, zi)) : zi;
        const float upper = (i < S - 1) ? nh_mul(0.5f, nh_add(zi, zr)) : zi;
        const float pr = nh_mul(perturb, prand[idx]);
        zi = nh_add(lower, nh_mul(nh_sub(upper, lower), pr));
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
    for (int j = lane; j < M; j += 64) part += (double)nh_add(wload(j), eps);
    const float total = (float)wave_sum(part);
    // cdf = [0, cumsum(pdf)]                                                      :31-33
    double carry = 0.0;
    for (int j0 = 0; j0 < M; j0 += 64) {
        const int j = j0 + lane;
        const float pdf = (j < M) ? nh_div(nh_add(wload(j), eps), total) : 0.0f;
        const double incl = wave_incl_sum((double)pdf, lane) + carry;
        if (j < M) cdf_s[j + 1] = (float)incl;
        carry = __shfl(incl, 63, 64);
    }
    if (lane == 0) cdf_s[0] = 0.0f;
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float invert_cdf(const float* cdf_s, const float* bins_s, int M, float u, float eps,
                                            int* ind_out = nullptr) {
    const int ind = upper_bound(cdf_s, M + 1, u);   // searchsorted(cdf,u,'right')   :42
    if (ind_out) *ind_out = ind;
    const int below = max(ind - 1, 0);              // :43
    const int above = min(ind, M);                  // :44
    const float cb = cdf_s[below], ca = cdf_s[above];
    const float bb = bins_s[below], ba = bins_s[above];
    float denom = nh_sub(ca, cb);
    if (denom < eps) denom = 1.0f;                  // :51
    const float t = nh_div(nh_sub(u, cb), denom);
    return nh_add(bb, nh_mul(t, nh_sub(ba, bb)));  // :54
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ bins, int64_t bins_stride,
                                                          const float* __restrict__ weights, int64_t w_stride,
                                                          const float* __restrict__ u, int64_t u_stride,
                                                          float* __restrict__ samples, int64_t B, int M, int K,
                                                          float eps, float* __restrict__ cdf_out,
                                                          int64_t* __restrict__ inds_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= B) return;
    float* cdf_s = lds + (size_t)wave * 2 * (M + 1);
    float* bins_s = cdf_s + (M + 1);
    for (int j = lane; j <= M; j += 64) bins_s[j] = bins[r * bins_stride + j];
    const float* wrow = weights + r * w_stride;
    build_cdf_wave([&](int j) { return wrow[j]; }, M, eps, cdf_s, lane);
    if (cdf_out)
        for (int j = lane; j <= M; j += 64) cdf_out[r * (M + 1) + j] = cdf_s[j];
    for (int k = lane; k < K; k += 64) {
        const float uk = u ? u[r * u_stride + k] : linspace01(k, K);
        int ind;
        samples[r * K + k] = invert_cdf(cdf_s, bins_s, M, uk, eps, &ind);
        if (inds_out) inds_out[r * K + k] = (int64_t)ind;
    }
}

// z_fine = sort(cat(z_coarse, sample_pdf(z_mid, w[:,1:-1], N_i)))        rendering.py:223-229
// Merge without a sort network: the coarse depths are already non-decreasing, so
//   rank(coarse i) = i + #{new < zc_i}            = i + prefix-sum over j<=i of hist[j],  hist[ub_k]++
//   rank(new k)    = ub_k + #{new q before k}      with ub_k = #{coarse <= zn_k} (binary search)
// and only the new-vs-new count is quadratic (N^2/64 compares per lane, LDS reads 16 B wide, broadcast).
__device__ __forceinline__ int fine_z_lds_floats(int S, int N) {
    const int S4 = (S + 3) & ~3, N4 = (N + 3) & ~3;
    return 3 * S4 + N4 + ((S + 1 + 3) & ~3) + N4;   // zc | cdf | bins | zn | hist | ub
}
__global__ __launch_bounds__(256) void fine_z_kernel(const float* __restrict__ zc, const float* __restrict__ wc,
                                                      const float* __restrict__ u, int64_t u_stride,
                                                      float* __restrict__ zf, float* __restrict__ znew_out,
                                                      int64_t B, int S, int N, float eps, float* __restrict__ cdf_out,
                                                      int64_t* __restrict__ inds_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= B) return;
    const int M = S - 2;                        // number of pdf bins
    const int S4 = (S + 3) & ~3, N4 = (N + 3) & ~3, H4 = (S + 1 + 3) & ~3;
    float* zc_s = lds + (size_t)wave * fine_z_lds_floats(S, N);
    float* cdf_s = zc_s + S4;                   // M+1 = S-1
    float* bins_s = cdf_s + S4;                 // M+1 = S-1 midpoints
    float* zn_s = bins_s + S4;                  // N new samples, padded with +inf to a multiple of 4
    int* hist_s = reinterpret_cast<int*>(zn_s + N4);   // S+1 counters
    int* ub_s = hist_s + H4;                    // N upper bounds
    for (int j = lane; j < S; j += 64) zc_s[j] = zc[r * S + j];
    for (int j = lane; j <= S; j += 64) hist_s[j] = 0;
    if (lane < N4 - N) zn_s[N + lane] = __builtin_inff();
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < S - 1; j += 64) bins_s[j] = nh_mul(0.5f, nh_add(zc_s[j], zc_s[j + 1]));  // :223
    const float* wrow = wc + r * S + 1;         // weights_coarse[:, 1:-1]          :225
    build_cdf_wave([&](int j) { return wrow[j]; }, M, eps, cdf_s, lane);
    if (cdf_out)
        for (int j = lane; j <= M; j += 64) cdf_out[r * (M + 1) + j] = cdf_s[j];
    for (int k = lane; k < N; k += 64) {
        const float uk = u ? u[r * u_stride + k] : linspace01(k, N);
        int ind;
        const float v = invert_cdf(cdf_s, bins_s, M, uk, eps, &ind);
        if (inds_out) inds_out[r * N + k] = (int64_t)ind;
        zn_s[k] = v;
        if (znew_out) znew_out[r * N + k] = v;
        const int ub = upper_bound(zc_s, S, v);  // #coarse <= v
        ub_s[k] = ub;
        atomicAdd(&hist_s[ub], 1);
    }
    __builtin_amdgcn_wave_barrier();
    float* out = zf + r * (S + N);
    // coarse elements: inclusive prefix of hist
    int carry = 0;
    for (int i0 = 0; i0 < S; i0 += 64) {
        const int i = i0 + lane;
        int c = (i < S) ? hist_s[i] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(c, o, 64);
            if (lane >= o) c += t;
        }
        c += carry;
        if (i < S) out[i + c] = zc_s[i];
        carry = __shfl(c, 63, 64);
    }
    // new elements: stable rank among the new ones (NaNs are not ordered)
    const float4* zn4 = reinterpret_cast<const float4*>(zn_s);
    for (int k = lane; k < N; k += 64) {
        const float x = zn_s[k];
        int rank = ub_s[k];
#pragma unroll 4
        for (int q4 = 0; q4 < N4 / 4; ++q4) {
            const float4 y = zn4[q4];
            const int q = q4 * 4;
            rank += (y.x < x || (y.x == x && q + 0 < k)) ? 1 : 0;
            rank += (y.y < x || (y.y == x && q + 1 < k)) ? 1 : 0;
            rank += (y.z < x || (y.z == x && q + 2 < k)) ? 1 : 0;
            rank += (y.w < x || (y.w == x && q + 3 < k)) ? 1 : 0;
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

extern "C" int nerfhip_sample_pdf_ex(const float* bins, int64_t bins_stride, const float* weights, int64_t w_stride,
                                     const float* u, int64_t u_stride, float* samples, int64_t B, int M, int K,
                                     float eps, float* cdf_out, int64_t* inds_out, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && M >= 1 && K >= 0 && M <= 2040);
    if (B == 0 || K == 0) return 0;
    NERFHIP_CHECK_ARG(bins && weights && samples);
    hipLaunchKernelGGL(nerfhip::sample_pdf_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256),
                       (size_t)4 * 2 * (M + 1) * sizeof(float), (hipStream_t)stream, bins, bins_stride, weights,
                       w_stride, u, u_stride, samples, B, M, K, eps, cdf_out, inds_out);
    return nerfhip_launch_status();
}
extern "C" int nerfhip_sample_pdf(const float* bins, int64_t bins_stride, const float* weights, int64_t w_stride,
                                  const float* u, int64_t u_stride, float* samples, int64_t B, int M, int K,
                                  float eps, nerfhip_stream_t stream) {
    return nerfhip_sample_pdf_ex(bins, bins_stride, weights, w_stride, u, u_stride, samples, B, M, K, eps, nullptr, nullptr,
                                 stream);
}

extern "C" int nerfhip_fine_z_ex(const float* z_coarse, const float* w_coarse, const float* u, int64_t u_stride,
                                 float* z_fine, float* z_new, int64_t B, int S_c, int N_i, float eps, float* cdf_out,
                                 int64_t* inds_out, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S_c >= 3 && N_i >= 1);
    const int S4 = (S_c + 3) & ~3, N4 = (N_i + 3) & ~3;
    const size_t per_wave = (size_t)(3 * S4 + N4 + ((S_c + 1 + 3) & ~3) + N4) * sizeof(float);
    NERFHIP_CHECK_ARG(4 * per_wave <= 65536);
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(z_coarse && w_coarse && z_fine);
    hipLaunchKernelGGL(nerfhip::fine_z_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 4 * per_wave,
                       (hipStream_t)stream, z_coarse, w_coarse, u, u_stride, z_fine, z_new, B, S_c, N_i, eps, cdf_out, inds_out);
    return nerfhip_launch_status();
}
extern "C" int nerfhip_fine_z(const float* z_coarse, const float* w_coarse, const float* u, int64_t u_stride,
                              float* z_fine, float* z_new, int64_t B, int S_c, int N_i, float eps,
                              nerfhip_stream_t stream) {
    return nerfhip_fine_z_ex(z_coarse, w_coarse, u, u_stride, z_fine, z_new, B, S_c, N_i, eps, nullptr, nullptr, stream);
}
