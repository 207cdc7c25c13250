// Ray geometry of reference datasets/ray_utils.py, shared by the ray kernels (rays.hip) and the batch-drawing kernel
// (draws.hip):   cam_dir : ray_utils.py:5-24     world_dir : ray_utils.py:27-52     ndc : ray_utils.py:55-94
#pragma once
#include "common.h"

namespace nerfhip {

struct Cam {
    float r[9];   // c2w[:, :3] row-major
    float t[3];   // c2w[:, 3]
};

__device__ __forceinline__ void cam_dir(int i, int j, int H, int W, float focal, float (&d)[3]) {
    // (i - W/2)/focal, -(j - H/2)/focal, -1     ray_utils.py:21-22  (W/2, H/2 are Python true divisions)
    d[0] = nh_div(nh_sub((float)i, (float)W * 0.5f), focal);
    d[1] = -nh_div(nh_sub((float)j, (float)H * 0.5f), focal);
    d[2] = -1.0f;
}

__device__ __forceinline__ void world_dir(const float (&d)[3], const float* __restrict__ c2w, float (&o)[3], float (&w)[3]) {
    // rays_d = directions @ c2w[:, :3].T : row k of c2w dotted with d, then normalised   ray_utils.py:43-44
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w[k] = nh_add(nh_add(nh_mul(d[0], c2w[4 * k]), nh_mul(d[1], c2w[4 * k + 1])), nh_mul(d[2], c2w[4 * k + 2]));
        o[k] = c2w[4 * k + 3];                                                          // ray_utils.py:46
    }
    const float n = sqrtf(nh_add(nh_add(nh_mul(w[0], w[0]), nh_mul(w[1], w[1])), nh_mul(w[2], w[2])));
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] = nh_div(w[k], n);
}

// sx = -1/(W/(2 focal)), sy = -1/(H/(2 focal)): Python-float (double) scalars in the reference, rounded to fp32 when
// they meet the tensors — computed on the host in double (ndc_scale) for bit parity.
__device__ __forceinline__ void ndc(float sx, float sy, float near, float (&o)[3], float (&d)[3]) {
    // ray_utils.py:76-92
    const float t = nh_div(-nh_add(near, o[2]), d[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = nh_add(o[k], nh_mul(t, d[k]));
    const float ox_oz = nh_div(o[0], o[2]), oy_oz = nh_div(o[1], o[2]);
    const float o0 = nh_mul(sx, ox_oz), o1 = nh_mul(sy, oy_oz);
    const float o2 = nh_add(1.0f, nh_div(nh_mul(2.0f, near), o[2]));
    const float d0 = nh_mul(sx, nh_sub(nh_div(d[0], d[2]), ox_oz));
    const float d1 = nh_mul(sy, nh_sub(nh_div(d[1], d[2]), oy_oz));
    o[0] = o0; o[1] = o1; o[2] = o2;
    d[0] = d0; d[1] = d1; d[2] = nh_sub(1.0f, o2);
}

// one ray (o, d, near, far) of global pixel id `id` (image * H*W + row * W + col) under the pose table c2w_all (n_images, 3, 4):
// the three functions above + the [o d near far] packing of blender.py:64-69 / llff.py:236-253
__device__ __forceinline__ void gen_ray(const float* __restrict__ c2w_all, int64_t id, int H, int W, float focal, float near,
                                        float far, int use_ndc, float ndc_plane, float sx, float sy, float* __restrict__ ray8) {
    const int64_t hw = (int64_t)H * W;
    const int64_t img = id / hw, q = id - img * hw;
    float d[3], o[3], w[3];
    cam_dir((int)(q % W), (int)(q / W), H, W, focal, d);
    world_dir(d, c2w_all + img * 12, o, w);
    if (use_ndc) ndc(sx, sy, ndc_plane, o, w);
    float4* out = reinterpret_cast<float4*>(ray8);
    out[0] = make_float4(o[0], o[1], o[2], w[0]);
    out[1] = make_float4(w[1], w[2], near, far);
}

}  // namespace nerfhip

// sx = -1/(W/(2 focal)), sy = -1/(H/(2 focal)) of ray_utils.py:84-85: Python doubles, rounded to fp32 when they meet the tensors
static inline float nerfhip_ndc_scale(int extent, double focal) { return (float)(-1.0 / ((double)extent / (2.0 * focal))); }
