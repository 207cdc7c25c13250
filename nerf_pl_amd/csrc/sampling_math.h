// Coarse depth sampling of reference models/rendering.py:183-204, shared by sample_coarse_z_kernel (sampling.hip) and the fused
// MLP forward, which can form its own coarse depths in the prologue (mlp_fwd_kernel.h, SURVEY section 2 "K0 fused into K2").
#pragma once
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

// linspace01 with the step formed once by the caller (three samples share it)
__device__ __forceinline__ float linspace01_step(int i, int S, float step) {
    if (S <= 1) return 0.0f;
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

// z of sample i of a ray with bounds (near, far): linear in depth or disparity (:189-193) and, when perturb > 0, jittered
// between the mid-points to its neighbours with the caller's U[0,1) draw `prand` (:197-204).  Same expressions as coarse_z_raw
// for samples i-1, i, i+1 with the shared terms (step, 1/near, 1/far) formed once: the divisions are IEEE-correct sequences of
// ~40 instructions each, and this runs inside the MLP kernels' prologue (code bytes against a 64 KiB instruction cache).
__device__ __forceinline__ float coarse_z_sample(float near, float far, int i, int S, int use_disp, float perturb, float prand) {
    const float step = (S > 1) ? nh_div(1.0f, (float)(S - 1)) : 0.0f;
    const float bn = use_disp ? nh_div(1.0f, near) : near, bf = use_disp ? nh_div(1.0f, far) : far;
    auto raw = [&](int k) {
        const float t = linspace01_step(k, S, step);
        const float v = nh_add(nh_mul(bn, nh_sub(1.0f, t)), nh_mul(bf, t));
        return use_disp ? nh_div(1.0f, v) : v;
    };
    float zi = raw(i);
    if (perturb > 0.0f) {
        const float zl = (i > 0) ? raw(i - 1) : zi;
        const float zr = (i < S - 1) ? raw(i + 1) : zi;
        const float lower = (i > 0) ? nh_mul(0.5f, nh_add(zl, zi)) : zi;
        const float upper = (i < S - 1) ? nh_mul(0.5f, nh_add(zi, zr)) : zi;
        const float pr = nh_mul(perturb, prand);
        zi = nh_add(lower, nh_mul(nh_sub(upper, lower), pr));
    }
    return zi;
}

}  // namespace nerfhip
