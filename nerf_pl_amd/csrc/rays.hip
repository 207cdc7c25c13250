// N1: the step before the path — ray generation of reference datasets/ray_utils.py on the device.
// The reference precomputes ALL rays of a dataset on the CPU (blender.py:42-69: 100 x H x W x 8 fp32 = 2 GB at 800^2)
// and ships 1024-row batches through a DataLoader; a ray is a pure function of (pixel, camera pose, focal), so here it
// is regenerated on the fly from an 8-byte pixel id: 8 B in, 32 B out per ray, HBM/latency-bound elementwise work.
//   ray_directions : ray_utils.py:5-24     get_rays : ray_utils.py:27-52     ndc_rays : ray_utils.py:55-94
//   gen_rays       : the three fused + the [o d near far] packing of blender.py:64-69 / llff.py:236-253
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

__global__ __launch_bounds__(256) void ray_directions_kernel(float* __restrict__ dirs, int H, int W, float focal) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    float d[3];
    cam_dir((int)(p % W), (int)(p / W), H, W, focal, d);
    dirs[3 * p] = d[0]; dirs[3 * p + 1] = d[1]; dirs[3 * p + 2] = d[2];
}

__global__ __launch_bounds__(256) void get_rays_kernel(const float* __restrict__ dirs, const float* __restrict__ c2w,
                                                        float* __restrict__ rays_o, float* __restrict__ rays_d, int64_t n) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float d[3] = {dirs[3 * p], dirs[3 * p + 1], dirs[3 * p + 2]};
    float o[3], w[3];
    world_dir(d, c2w, o, w);
#pragma unroll
    for (int k = 0; k < 3; ++k) { rays_o[3 * p + k] = o[k]; rays_d[3 * p + k] = w[k]; }
}

__global__ __launch_bounds__(256) void ndc_rays_kernel(float sx, float sy, float near, const float* __restrict__ ro,
                                                        const float* __restrict__ rd, float* __restrict__ out_o,
                                                        float* __restrict__ out_d, int64_t n) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float o[3] = {ro[3 * p], ro[3 * p + 1], ro[3 * p + 2]}, d[3] = {rd[3 * p], rd[3 * p + 1], rd[3 * p + 2]};
    ndc(sx, sy, near, o, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) { out_o[3 * p + k] = o[k]; out_d[3 * p + k] = d[k]; }
}

// pix: global pixel ids (image * H*W + row * W + col) or NULL = ids first_pix .. first_pix + n - 1
// rgbs_all / rgbs_out (both or neither): the batch's target colours rgbs_all[id] gathered in the same launch (blender.py:81-84)
__global__ __launch_bounds__(256) void gen_rays_kernel(const float* __restrict__ c2w_all, const int64_t* __restrict__ pix,
                                                        int64_t first_pix, int64_t n, int H, int W, float focal, float near,
                                                        float far, int use_ndc, float ndc_plane, float sx, float sy,
                                                        float* __restrict__ rays, const float* __restrict__ rgbs_all,
                                                        float* __restrict__ rgbs_out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int64_t id = pix ? pix[r] : first_pix + r;
    if (rgbs_out) {
        rgbs_out[3 * r] = rgbs_all[3 * id];
        rgbs_out[3 * r + 1] = rgbs_all[3 * id + 1];
        rgbs_out[3 * r + 2] = rgbs_all[3 * id + 2];
    }
    const int64_t hw = (int64_t)H * W;
    const int64_t img = id / hw, q = id - img * hw;
    float d[3], o[3], w[3];
    cam_dir((int)(q % W), (int)(q / W), H, W, focal, d);
    world_dir(d, c2w_all + img * 12, o, w);
    if (use_ndc) ndc(sx, sy, ndc_plane, o, w);
    float4* out = reinterpret_cast<float4*>(rays + r * 8);
    out[0] = make_float4(o[0], o[1], o[2], w[0]);
    out[1] = make_float4(w[1], w[2], near, far);
}

}  // namespace nerfhip

static inline float ndc_scale(int extent, double focal) { return (float)(-1.0 / ((double)extent / (2.0 * focal))); }

extern "C" int nerfhip_ray_directions(float* dirs, int H, int W, double focal, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(H >= 0 && W >= 0);
    const int64_t n = (int64_t)H * W;
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(dirs);
    hipLaunchKernelGGL(nerfhip::ray_directions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dirs, H,
                       W, (float)focal);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_get_rays(const float* directions, const float* c2w, float* rays_o, float* rays_d, int64_t n,
                                nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0);
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(directions && c2w && rays_o && rays_d);
    hipLaunchKernelGGL(nerfhip::get_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, directions,
                       c2w, rays_o, rays_d, n);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_ndc_rays(int H, int W, double focal, float near, const float* rays_o, const float* rays_d, float* out_o,
                                float* out_d, int64_t n, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0);
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(rays_o && rays_d && out_o && out_d);
    hipLaunchKernelGGL(nerfhip::ndc_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ndc_scale(W, focal), ndc_scale(H, focal), near, rays_o, rays_d, out_o, out_d, n);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_gen_rays(const float* c2w, const int64_t* pixel_ids, int64_t first_pixel, int64_t n, int H, int W,
                                double focal, float near, float far, int use_ndc, float ndc_near_plane, float* rays,
                                nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0 && H > 0 && W > 0);
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(c2w && rays);
    if (((uintptr_t)rays) & 15) return NERFHIP_E_ALIGN;
    hipLaunchKernelGGL(nerfhip::gen_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c2w,
                       pixel_ids, first_pixel, n, H, W, (float)focal, near, far, use_ndc, ndc_near_plane, ndc_scale(W, focal),
                       ndc_scale(H, focal), rays, (const float*)nullptr, (float*)nullptr);
    return nerfhip_launch_status();
}

extern "C" int nerfhip_sample_batch(const float* c2w, const int64_t* pixel_ids, const float* rgbs_all, int64_t n, int H, int W,
                                    double focal, float near, float far, int use_ndc, float ndc_near_plane, float* rays,
                                    float* rgbs, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0 && H > 0 && W > 0);
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(c2w && pixel_ids && rgbs_all && rays && rgbs);
    if (((uintptr_t)rays) & 15) return NERFHIP_E_ALIGN;
    hipLaunchKernelGGL(nerfhip::gen_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c2w,
                       pixel_ids, (int64_t)0, n, H, W, (float)focal, near, far, use_ndc, ndc_near_plane, ndc_scale(W, focal),
                       ndc_scale(H, focal), rays, rgbs_all, rgbs);
    return nerfhip_launch_status();
}
