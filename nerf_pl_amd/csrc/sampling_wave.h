// Per-wave inverse-CDF sampling machinery of reference models/rendering.py:14-55, 223-229 (one ray per 64-lane wavefront, the
// ray's cdf / bins staged in LDS), shared by the sampling kernels (sampling.hip) and the compositing kernel that appends the
// fine-pass depth assembly to the coarse pass's quadrature (composite.hip).
#pragma once
#include "sampling_math.h"

namespace nerfhip {

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

// ---- shared per-wave inverse-CDF machinery ---------------------------------------------------
// Row total of the pdf normaliser (rendering.py:30, `torch.sum(weights, -1, keepdim=True)` on fp32), two selectable roundings:
//   ROW_TOTAL_EXACT  the fp64 sum of the fp32 terms rounded once — the correctly rounded total (exact here: the terms span < 29
//                    binades), independent of any host; the default ARGUMENT of the wave helpers below and what the C entry points
//                    without a row_total parameter use (the Python operators pass ROW_TOTAL_ATEN by default: ops.set_row_total)
//   ROW_TOTAL_ATEN   the value ATen's CPU kernel returns, bit for bit: its fp32 additions in its own order (SumKernel.cpp
//                    cascade_sum -> vectorized_inner_sum -> row_sum: 8-float vectors — on every x86 capability of torch 2.x,
//                    AVX-512 builds included — four interleaved vector accumulators ("ILP") with a 16-step cascade, the
//                    leftover vectors into accumulator 0, accumulators 1-3 added to 0, then a scalar chain: 0 + the row's
//                    tail elements + the 8 vector lanes in order; rows shorter than 8 take the same path with 1-float
//                    "vectors").  The reference's searchsorted indices have knife edges on the last bit of this total
//                    (u == 1.0, cdf ties), so THIS is the mode that reproduces the (cdf, u) -> inds triples recorded at
//                    rendering.py:42 on 100 % of the elements (tests/test_gpu_parity.py); oracle/nerf_oracle.py
//                    aten_row_total is the same restatement in numpy, pinned against torch.sum itself.
constexpr int ROW_TOTAL_EXACT = 0, ROW_TOTAL_ATEN = 1;

// ATen multi_row_sum for ONE vector lane: `n` steps over four interleaved accumulators, step i adding term(4 i + k) to k
template <typename Term>
__device__ __forceinline__ void aten_ilp_cascade(Term term, int n, float (&ps)[4]) {
    int cl2 = 0;
    while ((1 << cl2) < n) ++cl2;                                   // utils::CeilLog2(n)
    const int level_power = (cl2 / 4 > 4) ? cl2 / 4 : 4;
    const int level_step = 1 << level_power, level_mask = level_step - 1;
    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[j][k] = 0.0f;
    int i = 0;
    while (i + level_step <= n) {
        for (int j = 0; j < level_step; ++j, ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[0][k] = nh_add(acc[0][k], term(4 * i + k));
#pragma unroll
        for (int j = 1; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[j][k] = nh_add(acc[j][k], acc[j - 1][k]); acc[j - 1][k] = 0.0f; }
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < n; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] = nh_add(acc[0][k], term(4 * i + k));
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] = nh_add(acc[0][k], acc[j][k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) ps[k] = acc[0][k];
}

// term(j), j in [0, M): the row's fp32 terms; every lane returns the total
template <typename Term>
__device__ __forceinline__ float aten_row_total_wave(Term term, int M, int lane) {
    const int V = (M >= 8) ? 8 : 1;                 // Vectorized<float> of the sum kernel; scalar path below one vector
    const int vs = M / V, n_ilp = vs / 4;
    float part = 0.0f;                              // lane l < V: vector lane l of row_sum's result
    if (lane < V) {
        auto vterm = [&](int i) { return term(i * V + lane); };
        float ps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (n_ilp > 0) aten_ilp_cascade(vterm, n_ilp, ps);
        for (int i = n_ilp * 4; i < vs; ++i) ps[0] = nh_add(ps[0], vterm(i));
        part = nh_add(nh_add(nh_add(ps[0], ps[1]), ps[2]), ps[3]);
    }
    float acc = 0.0f;
    if (V > 1)
        for (int k = vs * V; k < M; ++k) acc = nh_add(acc, term(k));            // the row's tail, scalar
    for (int k = 0; k < V; ++k) acc = nh_add(acc, __shfl(part, k, 64));         // then the vector lanes in order
    return acc;
}

// cdf_s: M+1 floats in LDS (built here), bins_s: M+1 floats in LDS (filled by the caller).
template <typename WLoad>
__device__ __forceinline__ void build_cdf_wave(WLoad wload, int M, float eps, float* cdf_s, int lane, int row_total = ROW_TOTAL_EXACT) {
    // weights + eps, total                                                        rendering.py:29-30
    float total;
    if (row_total == ROW_TOTAL_ATEN) {              // (wave-uniform)
        total = aten_row_total_wave([&](int j) { return nh_add(wload(j), eps); }, M, lane);
    } else {
        double part = 0.0;
        for (int j = lane; j < M; j += 64) part += (double)nh_add(wload(j), eps);
        total = (float)wave_sum(part);
    }
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

// z_fine = sort(cat(z_coarse, sample_pdf(z_mid, w[:,1:-1], N_i)))        rendering.py:223-229
// Merge without a sort network: the coarse depths are already non-decreasing, so
//   rank(coarse i) = i + #{new < zc_i}            = i + prefix-sum over j<=i of hist[j],  hist[ub_k]++
//   rank(new k)    = ub_k + #{new q before k}      with ub_k = #{coarse <= zn_k} (binary search)
// and only the new-vs-new count is quadratic (N^2/64 compares per lane, LDS reads 16 B wide, broadcast).
__device__ __forceinline__ int fine_z_lds_floats(int S, int N) {
    const int S4 = (S + 3) & ~3, N4 = (N + 3) & ~3;
    return 3 * S4 + N4 + ((S + 1 + 3) & ~3) + N4;   // zc | cdf | bins | zn | hist | ub
}
// one ray per wave: `lds` = this wave's fine_z_lds_floats(S, N) floats; zrow = the ray's S coarse depths (global), wload(j) = its
// coarse weight 1 + j (weights_coarse[:, 1:-1], :225: from global memory or from the LDS of the kernel that just composited them),
// urow = its N uniforms or NULL (deterministic linspace); out = its S + N fine depths; znew / cdf_row / inds_row optional exports
template <typename WLoad>
__device__ __forceinline__ void fine_z_wave(float* lds, const float* __restrict__ zrow, WLoad wload, const float* __restrict__ urow,
                                            int S, int N, float eps, float* __restrict__ out, float* __restrict__ znew,
                                            float* __restrict__ cdf_row, int64_t* __restrict__ inds_row, int lane,
                                            int row_total = ROW_TOTAL_EXACT) {
    const int M = S - 2;                        // number of pdf bins
    const int S4 = (S + 3) & ~3, N4 = (N + 3) & ~3, H4 = (S + 1 + 3) & ~3;
    float* zc_s = lds;
    float* cdf_s = zc_s + S4;                   // M+1 = S-1
    float* bins_s = cdf_s + S4;                 // M+1 = S-1 midpoints
    float* zn_s = bins_s + S4;                  // N new samples, padded with +inf to a multiple of 4
    int* hist_s = reinterpret_cast<int*>(zn_s + N4);   // S+1 counters
    int* ub_s = hist_s + H4;                    // N upper bounds
    for (int j = lane; j < S; j += 64) zc_s[j] = zrow[j];
    for (int j = lane; j <= S; j += 64) hist_s[j] = 0;
    if (lane < N4 - N) zn_s[N + lane] = __builtin_inff();
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < S - 1; j += 64) bins_s[j] = nh_mul(0.5f, nh_add(zc_s[j], zc_s[j + 1]));  // :223
    build_cdf_wave(wload, M, eps, cdf_s, lane, row_total);
    if (cdf_row)
        for (int j = lane; j <= M; j += 64) cdf_row[j] = cdf_s[j];
    for (int k = lane; k < N; k += 64) {
        const float uk = urow ? urow[k] : linspace01(k, N);
        int ind;
        const float v = invert_cdf(cdf_s, bins_s, M, uk, eps, &ind);
        if (inds_row) inds_row[k] = (int64_t)ind;
        zn_s[k] = v;
        if (znew) znew[k] = v;
        const int ub = upper_bound(zc_s, S, v);  // #coarse <= v
        ub_s[k] = ub;
        atomicAdd(&hist_s[ub], 1);
    }
    __builtin_amdgcn_wave_barrier();
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
    // new elements: stable rank among the new ones (NaNs are not ordered).  The inverse cdf is monotone, so with the deterministic
    // uniforms of a test_time render (rendering.py:36-37, every eval chunk) the new samples arrive already sorted: then the rank
    // among them IS the index — checked, not assumed (fp32 rounding of the lerp at a bin edge could in principle invert a pair),
    // and the N^2 / 64 comparisons per lane below (half of this wave's instructions at N = 128) are skipped.
    bool in_order = true;
    for (int k = lane; k < N; k += 64) in_order = in_order && (k == 0 || zn_s[k - 1] <= zn_s[k]);
    if (__all(in_order)) {                  // (wave-uniform)
        for (int k = lane; k < N; k += 64) out[ub_s[k] + k] = zn_s[k];
        return;
    }
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
