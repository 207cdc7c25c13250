// Per-wave alpha compositing of reference models/rendering.py:143-172 (one ray per 64-lane wavefront) and its backward: the
// device code shared by the compositing kernels (composite.hip) and the single-launch render kernels, whose workgroups composite
// their own rays once the MLP sub-passes have produced them (mlp_render_kernel.h).  Transmittance is a wave-wide exclusive prefix
// product (fp64, rounded per element like torch-CPU cumprod) chained across 64-sample chunks.
#pragma once
#include "loss_math.h"
#include "sampling_wave.h"

namespace nerfhip {

struct RayGeom {
    float dnorm;
};

__device__ __forceinline__ float ray_dnorm(const float* __restrict__ rays, int64_t r) {
    const float dx = rays[r * 8 + 3], dy = rays[r * 8 + 4], dz = rays[r * 8 + 5];
    return sqrtf(nh_add(nh_add(nh_mul(dx, dx), nh_mul(dy, dy)), nh_mul(dz, dz)));  // :150
}

// per-sample quantities shared by forward and backward
struct SampleTerms {
    float delta;  // (z[i+1]-z[i] | 1e10) * |d|
    float e;      // exp(-delta * relu(sigma+noise)) = 1 - alpha
    float alpha;
    float sh;     // (1 - alpha) + 1e-10, the factor entering the transmittance product
    bool on;      // relu gate open
};

__device__ __forceinline__ SampleTerms sample_terms(float z_i, float z_next, bool last, float dnorm, float sigma,
                                                    float noise) {
    SampleTerms t;
    const float d = last ? 1e10f : nh_sub(z_next, z_i);   // :144-146
    t.delta = nh_mul(d, dnorm);                            // :150
    const float s = nh_add(sigma, noise);
    t.on = s > 0.0f;
    const float sr = t.on ? s : 0.0f;                         // relu   :155
    t.e = expf(-nh_mul(t.delta, sr));
    t.alpha = nh_sub(1.0f, t.e);
    t.sh = nh_add(nh_sub(1.0f, t.alpha), 1e-10f);       // :157
    return t;
}

// Forward quadrature of one ray by one wave (rendering.py:143-172): weights (global, NULL ok) and / or w_s (LDS, NULL ok) receive
// w_i = alpha_i T_i; lane 0 stores opacity and, for RAW_CH == 4, rgb / depth (each pointer may be NULL: test_time keeps only the
// coarse opacity, rendering.py:209-213).
template <int RAW_CH>
__device__ __forceinline__ void composite_fwd_wave(const float* __restrict__ raw, const float* __restrict__ z,
                                                   const float* __restrict__ rays, const float* __restrict__ noise, float noise_std,
                                                   int white_back, float* __restrict__ weights, float* __restrict__ rgb,
                                                   float* __restrict__ depth, float* __restrict__ opacity, int64_t r, int S,
                                                   float* w_s, int lane) {
    const float dnorm = ray_dnorm(rays, r);
    const float* zr = z + r * S;
    double carry = 1.0;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_o = 0.f;
    for (int i0 = 0; i0 < S; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < S;
        float sigma = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, zi = 0.f, zn = 0.f, nz = 0.f;
        if (valid) {
            zi = zr[i];
            zn = (i + 1 < S) ? zr[i + 1] : zi;
            if (RAW_CH == 4) {
                const float4 v = reinterpret_cast<const float4*>(raw)[r * S + i];
                cr = v.x; cg = v.y; cb = v.z; sigma = v.w;
            } else {
                sigma = raw[r * S + i];
            }
            if (noise) nz = nh_mul(noise[r * S + i], noise_std);   // :152
        }
        const SampleTerms t = sample_terms(zi, zn, i == S - 1, dnorm, sigma, nz);
        const double f = valid ? (double)t.sh : 1.0;
        const double incl = wave_incl_prod(f, lane);
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        const float T = (float)(excl * carry);                        // cumprod(...)[:, :-1]  :158-159
        carry = carry * __shfl(incl, 63, 64);
        const float w = valid ? nh_mul(t.alpha, T) : 0.0f;
        if (valid) {
            if (weights) weights[r * S + i] = w;
            if (w_s) w_s[i] = w;
        }
        acc_o += w;
        if (RAW_CH == 4) {
            acc_r += w * cr; acc_g += w * cg; acc_b += w * cb;
            acc_d += w * zi;
        }
    }
    acc_o = wave_sum(acc_o);
    if (RAW_CH == 4) {
        acc_r = wave_sum(acc_r); acc_g = wave_sum(acc_g); acc_b = wave_sum(acc_b); acc_d = wave_sum(acc_d);
    }
    if (lane == 0) {
        if (opacity) opacity[r] = acc_o;                              // weights.sum(1)  :160
        if (RAW_CH == 4) {
            const float bg = white_back ? nh_sub(1.0f, acc_o) : 0.0f;   // :169-170
            if (rgb) {
                rgb[r * 3 + 0] = acc_r + bg;
                rgb[r * 3 + 1] = acc_g + bg;
                rgb[r * 3 + 2] = acc_b + bg;
            }
            if (depth) depth[r] = acc_d;                               // :167
        }
    }
}

// Training fast path ("fused loss + PSNR + composite backward seed"): the quadrature of composite_fwd_kernel<4>,
// the MSE residual of the ray against its target colour (losses.py:9-14: d loss / d rgb = 2 (rgb - t) / n, `gscale` = 2 / n) and
// the backward of composite_bwd_kernel<4> for exactly that upstream gradient, in ONE launch — three launches (forward, loss
// gradient scaling, backward) and an HBM round trip of rgb fewer per pass.  Every value is formed by the same expressions in
// the same order as in the separate kernels, so g_raw is bit-identical to composite_fwd -> mse_psnr -> composite_bwd.
// One ray per wave.  T_s: S floats of LDS (transmittance between the sweeps); w_s (NULL ok): S floats of LDS that receive the
// ray's weights for a consumer in the same kernel (the fine-pass depth assembly below) instead of / besides `weights` in HBM.
template <bool RGB_THROUGH = false>
__device__ __forceinline__ void composite_train_wave(const float* __restrict__ raw, const float* __restrict__ z,
                                                     const float* __restrict__ rays, const float* __restrict__ noise,
                                                     float noise_std, int white_back, const float* __restrict__ target,
                                                     float gscale, float* __restrict__ weights, float* __restrict__ rgb,
                                                     float* __restrict__ depth, float* __restrict__ opacity,
                                                     float* __restrict__ g_raw, int64_t r, int S, float* T_s, float* w_s, int lane) {
    const float dnorm = ray_dnorm(rays, r);
    const float* zr = z + r * S;
    // ---- forward sweep (composite_fwd_kernel<4>) ----
    double carry = 1.0;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_o = 0.f;
    for (int i0 = 0; i0 < S; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < S;
        float sigma = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, zi = 0.f, zn = 0.f, nz = 0.f;
        if (valid) {
            zi = zr[i];
            zn = (i + 1 < S) ? zr[i + 1] : zi;
            const float4 v = reinterpret_cast<const float4*>(raw)[r * S + i];
            cr = v.x; cg = v.y; cb = v.z; sigma = v.w;
            if (noise) nz = nh_mul(noise[r * S + i], noise_std);
        }
        const SampleTerms t = sample_terms(zi, zn, i == S - 1, dnorm, sigma, nz);
        const double f = valid ? (double)t.sh : 1.0;
        const double incl = wave_incl_prod(f, lane);
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        const float T = (float)(excl * carry);
        carry = carry * __shfl(incl, 63, 64);
        const float w = valid ? nh_mul(t.alpha, T) : 0.0f;
        if (valid) {
            if (weights) weights[r * S + i] = w;
            if (w_s) w_s[i] = w;
            T_s[i] = T;
        }
        acc_o += w;
        acc_r += w * cr; acc_g += w * cg; acc_b += w * cb;
        acc_d += w * zi;
    }
    acc_o = wave_sum(acc_o);
    acc_r = wave_sum(acc_r); acc_g = wave_sum(acc_g); acc_b = wave_sum(acc_b); acc_d = wave_sum(acc_d);
    const float bg = white_back ? nh_sub(1.0f, acc_o) : 0.0f;
    const float out_r = acc_r + bg, out_g = acc_g + bg, out_b = acc_b + bg;
    if (lane == 0) {
        opacity[r] = acc_o;
        if (RGB_THROUGH) {      // device-scope (write-through) stores: another workgroup of THIS launch reads the colours back
            __hip_atomic_store(rgb + r * 3 + 0, out_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(rgb + r * 3 + 1, out_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(rgb + r * 3 + 2, out_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            rgb[r * 3 + 0] = out_r;
            rgb[r * 3 + 1] = out_g;
            rgb[r * 3 + 2] = out_b;
        }
        depth[r] = acc_d;
    }
    // ---- d loss / d rgb of this ray (mse_psnr_kernel: (rgb - t) * (2 / n)) ----
    const float gr = nh_mul(nh_sub(out_r, target[r * 3]), gscale);
    const float gg = nh_mul(nh_sub(out_g, target[r * 3 + 1]), gscale);
    const float gb = nh_mul(nh_sub(out_b, target[r * 3 + 2]), gscale);
    const float gd = 0.f;
    float gconst = 0.f;
    if (white_back) gconst -= (gr + gg + gb);
    __builtin_amdgcn_wave_barrier();
    // ---- reverse sweep (composite_bwd_kernel<4>): exclusive suffix sum of gw_k*w_k, colour and density gradients ----
    float tail = 0.f;
    const int nchunk = (S + 63) / 64;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int i = c * 64 + lane;
        const bool valid = i < S;
        SampleTerms t{};
        float T = 0.f, gw = 0.f;
        if (valid) {
            const float zi = zr[i];
            const float zn = (i + 1 < S) ? zr[i + 1] : zi;
            const float4 v = reinterpret_cast<const float4*>(raw)[r * S + i];
            const float nz = noise ? nh_mul(noise[r * S + i], noise_std) : 0.f;
            t = sample_terms(zi, zn, i == S - 1, dnorm, v.w, nz);
            T = T_s[i];
            const float w = t.alpha * T;
            gw = gconst + 0.f;
            gw += gr * v.x + gg * v.y + gb * v.z + gd * zi;
            float* o = g_raw + (r * S + i) * 4;
            o[0] = w * gr; o[1] = w * gg; o[2] = w * gb;
        }
        const float vv = valid ? gw * (t.alpha * T) : 0.f;
        float incl = vv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float tt = __shfl_down(incl, o, 64);
            if (lane + o < 64) incl += tt;
        }
        const float suf = (incl - vv) + tail;
        tail += __shfl(incl, 0, 64);
        if (valid) {
            const float g_alpha = gw * T - suf / t.sh;
            const float g_sigma = t.on ? g_alpha * t.delta * t.e : 0.f;
            g_raw[(r * S + i) * 4 + 3] = g_sigma;
        }
    }
}

// the weights of one ray (the forward sweep above without the colours) into LDS: w_s[i] = alpha_i T_i, same expressions, same bits
// (RAW_CH 4: raw = [r g b sigma] per point; 1: the sigma-only coarse pass of test_time, rendering.py:209-213)
template <int RAW_CH = 4>
__device__ __forceinline__ void composite_weights_wave(const float* __restrict__ raw, const float* __restrict__ z,
                                                       const float* __restrict__ rays, const float* __restrict__ noise,
                                                       float noise_std, int64_t r, int S, float* w_s, int lane) {
    const float dnorm = ray_dnorm(rays, r);
    const float* zr = z + r * S;
    double carry = 1.0;
    for (int i0 = 0; i0 < S; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < S;
        float sigma = 0.f, zi = 0.f, zn = 0.f, nz = 0.f;
        if (valid) {
            zi = zr[i];
            zn = (i + 1 < S) ? zr[i + 1] : zi;
            sigma = RAW_CH == 4 ? raw[(r * S + i) * 4 + 3] : raw[r * S + i];
            if (noise) nz = nh_mul(noise[r * S + i], noise_std);
        }
        const SampleTerms t = sample_terms(zi, zn, i == S - 1, dnorm, sigma, nz);
        const double f = valid ? (double)t.sh : 1.0;
        const double incl = wave_incl_prod(f, lane);
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        const float T = (float)(excl * carry);
        carry = carry * __shfl(incl, 63, 64);
        if (valid) w_s[i] = nh_mul(t.alpha, T);
    }
}

}  // namespace nerfhip
