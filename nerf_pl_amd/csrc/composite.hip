// K4 / K4b: alpha compositing of one ray per 64-lane wavefront — replaces the ~15-launch
// sub/cat/mul/norm/relu/exp/cat/cumprod/mul/sum sequence of `inference` (reference
// models/rendering.py:143-172) and its autograd mirror.  HBM-bound: S*(16+4[+4]) B in and
// S*4 + 20 B out per ray, every byte touched once, loads coalesced along the sample axis.
// Transmittance is a wave-wide exclusive prefix product (fp64, rounded per element like
// torch-CPU cumprod) chained across 64-sample chunks.
#include "composite_wave.h"

namespace nerfhip {

template <int RAW_CH>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ raw,
                                                             const float* __restrict__ z,
                                                             const float* __restrict__ rays,
                                                             const float* __restrict__ noise, float noise_std,
                                                             int white_back, float* __restrict__ weights,
                                                             float* __restrict__ rgb, float* __restrict__ depth,
                                                             float* __restrict__ opacity, int64_t B, int S) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= B) return;
    composite_fwd_wave<RAW_CH>(raw, z, rays, noise, noise_std, white_back, weights, rgb, depth, opacity, r, S, nullptr, lane);
}

// Backward.  L depends on raw through  rgb = sum_i w_i c_i + wb*(1-sum w),  depth = sum w z,
// opacity = sum w, and (optionally) the weights themselves.  With gw_i = dL/dw_i:
//   dL/dc_i     = w_i * g_rgb
//   dL/dalpha_i = gw_i*T_i - (sum_{k>i} gw_k w_k) / sh_i         (T_k carries the factor sh_i for k>i)
//   dL/dsigma_i = dL/dalpha_i * delta_i * e_i * [sigma_i+noise_i > 0]
template <int RAW_CH>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ raw,
                                                             const float* __restrict__ z,
                                                             const float* __restrict__ rays,
                                                             const float* __restrict__ noise, float noise_std,
                                                             int white_back, const float* __restrict__ g_rgb,
                                                             const float* __restrict__ g_depth,
                                                             const float* __restrict__ g_opacity,
                                                             const float* __restrict__ g_weights,
                                                             float* __restrict__ g_raw, int64_t B, int S) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= B) return;
    float* T_s = lds + (size_t)wave * 2 * S;   // transmittance T_i
    float* gw_s = T_s + S;                     // dL/dw_i
    const float dnorm = ray_dnorm(rays, r);
    const float* zr = z + r * S;
    float gr = 0.f, gg = 0.f, gb = 0.f;
    if (RAW_CH == 4 && g_rgb) { gr = g_rgb[r * 3]; gg = g_rgb[r * 3 + 1]; gb = g_rgb[r * 3 + 2]; }
    const float gd = (RAW_CH == 4 && g_depth) ? g_depth[r] : 0.f;
    float gconst = g_opacity ? g_opacity[r] : 0.f;
    if (RAW_CH == 4 && white_back) gconst -= (gr + gg + gb);
    // forward sweep: transmittance + dL/dw + colour gradients
    double carry = 1.0;
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
            if (noise) nz = nh_mul(noise[r * S + i], noise_std);
        }
        const SampleTerms t = sample_terms(zi, zn, i == S - 1, dnorm, sigma, nz);
        const double f = valid ? (double)t.sh : 1.0;
        const double incl = wave_incl_prod(f, lane);
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        const float T = (float)(excl * carry);
        carry = carry * __shfl(incl, 63, 64);
        if (valid) {
            const float w = t.alpha * T;
            float gw = gconst + (g_weights ? g_weights[r * S + i] : 0.f);
            if (RAW_CH == 4) {
                gw += gr * cr + gg * cg + gb * cb + gd * zi;
                float* o = g_raw + (r * S + i) * 4;
                o[0] = w * gr; o[1] = w * gg; o[2] = w * gb;
            }
            T_s[i] = T;
            gw_s[i] = gw;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // reverse sweep: exclusive suffix sum of gw_k*w_k, then dL/dsigma
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
            const float sigma = (RAW_CH == 4) ? raw[(r * S + i) * 4 + 3] : raw[r * S + i];
            const float nz = noise ? nh_mul(noise[r * S + i], noise_std) : 0.f;
            t = sample_terms(zi, zn, i == S - 1, dnorm, sigma, nz);
            T = T_s[i];
            gw = gw_s[i];
        }
        const float v = valid ? gw * (t.alpha * T) : 0.f;
        float incl = v;   // inclusive suffix scan: lane l gets the sum over lanes >= l
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float tt = __shfl_down(incl, o, 64);
            if (lane + o < 64) incl += tt;
        }
        const float suf = (incl - v) + tail;   // strictly-after sum
        tail += __shfl(incl, 0, 64);
        if (valid) {
            const float g_alpha = gw * T - suf / t.sh;
            const float g_sigma = t.on ? g_alpha * t.delta * t.e : 0.f;
            g_raw[(r * S + i) * RAW_CH + (RAW_CH - 1)] = g_sigma;
        }
    }
}

__global__ __launch_bounds__(256) void composite_train_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                              const float* __restrict__ rays, const float* __restrict__ noise,
                                                              float noise_std, int white_back, const float* __restrict__ target,
                                                              float gscale, float* __restrict__ weights, float* __restrict__ rgb,
                                                              float* __restrict__ depth, float* __restrict__ opacity,
                                                              float* __restrict__ g_raw, int64_t B, int S) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= B) return;
    composite_train_wave(raw, z, rays, noise, noise_std, white_back, target, gscale, weights, rgb, depth, opacity, g_raw, r, S,
                         lds + (size_t)wave * S, nullptr, lane);
}

// The coarse pass of a training step: composite_train_kernel + the fine-pass depth assembly of rendering.py:223-229
// (fine_z_kernel of sampling.hip: z_mid, sample_pdf on weights[:, 1:-1], sort(cat)) for the same ray in the same wave — the
// weights go from the quadrature to the inverse-CDF sampling through LDS and need not exist in HBM at all (`weights` NULL).
// Same expressions in the same order as the two kernels: bit-identical z_fine.
__global__ __launch_bounds__(256) void composite_train_fine_z_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                                     const float* __restrict__ rays, const float* __restrict__ noise,
                                                                     float noise_std, int white_back, const float* __restrict__ target,
                                                                     float gscale, float* __restrict__ weights, float* __restrict__ rgb,
                                                                     float* __restrict__ depth, float* __restrict__ opacity,
                                                                     float* __restrict__ g_raw, int64_t B, int S,
                                                                     const float* __restrict__ u, int64_t u_stride, int N, float eps,
                                                                     float* __restrict__ z_fine, int row_total) {
    // Two rays per workgroup, two waves per ray — both chains are latency-bound and independent once the weights exist, so they
    // run side by side: waves 0-1 composite (quadrature, loss gradient, backward sweep), waves 2-3 form the same rays' weights
    // again (forward sweep only, raw is in L2) and assemble the fine depths from them.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 2 + (wave & 1);
    if (r >= B) return;
    const int S4 = (S + 3) & ~3;
    float* base = lds + (size_t)(wave & 1) * (2 * S4 + fine_z_lds_floats(S, N));
    if (wave < 2) {
        composite_train_wave(raw, z, rays, noise, noise_std, white_back, target, gscale, weights, rgb, depth, opacity, g_raw, r, S,
                             base, nullptr, lane);
    } else {
        float* w_s = base + S4;
        composite_weights_wave(raw, z, rays, noise, noise_std, r, S, w_s, lane);
        __builtin_amdgcn_wave_barrier();
        fine_z_wave(w_s + S4, z + r * S, [&](int j) { return w_s[1 + j]; }, u ? u + r * u_stride : nullptr, S, N, eps,
                    z_fine + r * (S + N), nullptr, nullptr, nullptr, lane, row_total);
    }
}

// The fine (last) pass of a training step: composite_train_kernel + the step's loss values (mse_psnr_kernel of loss.hip without
// its gradient outputs): every workgroup announces its rays' finished colours with an arrival ticket, and the last one to
// arrive reduces the two images against the target in mse_psnr_kernel's own order (loss_math.h) — out3 = [loss, psnr, mse],
// bit-identical to the separate launch.  `ticket` is a zero-initialised device word the kernel leaves at zero.
__global__ __launch_bounds__(256) void composite_train_loss_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                                   const float* __restrict__ rays, const float* __restrict__ noise,
                                                                   float noise_std, int white_back, const float* __restrict__ target,
                                                                   float gscale, float* __restrict__ weights, float* rgb,
                                                                   float* __restrict__ depth, float* __restrict__ opacity,
                                                                   float* __restrict__ g_raw, int64_t B, int S,
                                                                   const float* __restrict__ rgb_coarse, float* __restrict__ out3,
                                                                   unsigned* __restrict__ ticket) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[2][16];
    __shared__ unsigned last_s;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r < B)
        composite_train_wave<true>(raw, z, rays, noise, noise_std, white_back, target, gscale, weights, rgb, depth, opacity, g_raw, r, S,
                                   lds + (size_t)wave * S, nullptr, lane);
    // The colours left this wave as device-scope write-through stores; once they are acknowledged (vmcnt) they are visible to
    // every workgroup of the device, and the ticket below may be taken.  (A release FENCE here would write back this XCD's whole
    // L2 — 3 MB of g_raw — once per workgroup: measured 20 us for the launch instead of 7.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = (prev == gridDim.x - 1) ? 1u : 0u;
        if (prev == gridDim.x - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last_s) return;
    // consumer side of the hand-off: an agent-scope ACQUIRE in the one workgroup that reads the others' colours.  The
    // producer side stays write-through stores + s_waitcnt vmcnt(0) + the relaxed ticket: an agent-scope RELEASE there writes this
    // XCD's whole L2 back once per workgroup (measured: 20 us per launch instead of 7, round 4).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // the fine image written by the OTHER workgroups of this launch: device-scope loads (this XCD's L2 is not theirs)
    const float* rgb_f = rgb;
    auto fresh = [&](int64_t i) { return __hip_atomic_load(rgb_f + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (rgb_coarse)
        mse_psnr_block<4>([&](int64_t i) { return rgb_coarse[i]; }, fresh, true, target, 3 * B, out3, nullptr, nullptr, red);
    else                            // N_importance == 0: this pass IS the coarse pass, the loss has one term
        mse_psnr_block<4>(fresh, [&](int64_t i) { return 0.0f; }, false, target, 3 * B, out3, nullptr, nullptr, red);
}

}  // namespace nerfhip

extern "C" int nerfhip_composite_train(const float* raw, const float* z, const float* rays, const float* noise, float noise_std,
                                       int white_back, const float* target, float grad_scale, float* weights, float* rgb,
                                       float* depth, float* opacity, float* g_raw, int64_t B, int S, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S >= 1 && S <= 2048);
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(raw && z && rays && target && rgb && depth && opacity && g_raw);
    if ((((uintptr_t)raw) | ((uintptr_t)g_raw)) & 15) return NERFHIP_E_ALIGN;
    if (noise_std == 0.0f) noise = nullptr;
    hipLaunchKernelGGL(nerfhip::composite_train_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), (size_t)4 * S * sizeof(float),
                       (hipStream_t)stream, raw, z, rays, noise, noise_std, white_back, target, grad_scale, weights, rgb, depth,
                       opacity, g_raw, B, S);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_composite_fwd(const float* raw, int raw_ch, const float* z, const float* rays,
                                     const float* noise, float noise_std, int white_back, float* weights, float* rgb,
                                     float* depth, float* opacity, int64_t B, int S, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S >= 1 && (raw_ch == 1 || raw_ch == 4));
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(raw && z && rays && weights && opacity);
    if (raw_ch == 4) {
        NERFHIP_CHECK_ARG(rgb && depth);
        if (((uintptr_t)raw) & 15) return NERFHIP_E_ALIGN;
    }
    if (noise_std == 0.0f) noise = nullptr;
    dim3 grid((unsigned)((B + 3) / 4)), block(256);
    if (raw_ch == 4)
        hipLaunchKernelGGL(nerfhip::composite_fwd_kernel<4>, grid, block, 0, (hipStream_t)stream, raw, z, rays, noise,
                           noise_std, white_back, weights, rgb, depth, opacity, B, S);
    else
        hipLaunchKernelGGL(nerfhip::composite_fwd_kernel<1>, grid, block, 0, (hipStream_t)stream, raw, z, rays, noise,
                           noise_std, white_back, weights, rgb, depth, opacity, B, S);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_composite_bwd(const float* raw, int raw_ch, const float* z, const float* rays,
                                     const float* noise, float noise_std, int white_back, const float* g_rgb,
                                     const float* g_depth, const float* g_opacity, const float* g_weights,
                                     float* g_raw, int64_t B, int S, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S >= 1 && S <= 2048 && (raw_ch == 1 || raw_ch == 4));
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(raw && z && rays && g_raw);
    if (raw_ch == 4 && (((uintptr_t)raw) & 15)) return NERFHIP_E_ALIGN;
    if (noise_std == 0.0f) noise = nullptr;
    dim3 grid((unsigned)((B + 3) / 4)), block(256);
    size_t lds = (size_t)4 * 2 * S * sizeof(float);
    if (raw_ch == 4)
        hipLaunchKernelGGL(nerfhip::composite_bwd_kernel<4>, grid, block, lds, (hipStream_t)stream, raw, z, rays,
                           noise, noise_std, white_back, g_rgb, g_depth, g_opacity, g_weights, g_raw, B, S);
    else
        hipLaunchKernelGGL(nerfhip::composite_bwd_kernel<1>, grid, block, lds, (hipStream_t)stream, raw, z, rays,
                           noise, noise_std, white_back, g_rgb, g_depth, g_opacity, g_weights, g_raw, B, S);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_composite_train_fine_z(const float* raw, const float* z, const float* rays, const float* noise, float noise_std,
                                              int white_back, const float* target, float grad_scale, float* weights, float* rgb,
                                              float* depth, float* opacity, float* g_raw, int64_t B, int S, const float* u,
                                              int64_t u_stride, int N_i, float eps, float* z_fine, int row_total,
                                              nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S >= 3 && S <= 2048 && N_i >= 1 && (row_total == 0 || row_total == 1));
    const int S4 = (S + 3) & ~3, N4 = (N_i + 3) & ~3;
    const size_t per_wave = (size_t)(2 * S4 + 3 * S4 + N4 + ((S + 1 + 3) & ~3) + N4) * sizeof(float);
    NERFHIP_CHECK_ARG(2 * per_wave <= 65536);
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(raw && z && rays && target && rgb && depth && opacity && g_raw && z_fine);
    if ((((uintptr_t)raw) | ((uintptr_t)g_raw)) & 15) return NERFHIP_E_ALIGN;
    if (noise_std == 0.0f) noise = nullptr;
    hipLaunchKernelGGL(nerfhip::composite_train_fine_z_kernel, dim3((unsigned)((B + 1) / 2)), dim3(256), 2 * per_wave,
                       (hipStream_t)stream, raw, z, rays, noise, noise_std, white_back, target, grad_scale, weights, rgb, depth,
                       opacity, g_raw, B, S, u, u_stride, N_i, eps, z_fine, row_total);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_composite_train_loss(const float* raw, const float* z, const float* rays, const float* noise, float noise_std,
                                            int white_back, const float* target, float grad_scale, float* weights, float* rgb,
                                            float* depth, float* opacity, float* g_raw, int64_t B, int S, const float* rgb_coarse,
                                            float* out3, uint32_t* ticket, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S >= 1 && S <= 2048);
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(raw && z && rays && target && rgb && depth && opacity && g_raw && out3 && ticket);
    if ((((uintptr_t)raw) | ((uintptr_t)g_raw)) & 15) return NERFHIP_E_ALIGN;
    if (noise_std == 0.0f) noise = nullptr;
    hipLaunchKernelGGL(nerfhip::composite_train_loss_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), (size_t)4 * S * sizeof(float),
                       (hipStream_t)stream, raw, z, rays, noise, noise_std, white_back, target, grad_scale, weights, rgb, depth,
                       opacity, g_raw, B, S, rgb_coarse, out3, ticket);
    return nerfhip_launch_status();
}
