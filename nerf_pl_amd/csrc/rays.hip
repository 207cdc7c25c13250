// N1: the step before the path — ray generation of reference datasets/ray_utils.py on the device.
// The reference precomputes ALL rays of a dataset on the CPU (blender.py:42-69: 100 x H x W x 8 fp32 = 2 GB at 800^2)
// and ships 1024-row batches through a DataLoader; a ray is a pure function of (pixel, camera pose, focal), so here it
// is regenerated on the fly from an 8-byte pixel id: 8 B in, 32 B out per ray, HBM/latency-bound elementwise work.
//   ray_directions : ray_utils.py:5-24     get_rays : ray_utils.py:27-52     ndc_rays : ray_utils.py:55-94
//   gen_rays       : the three fused + the [o d near far] packing of blender.py:64-69 / llff.py:236-253
#include "rays_math.h"

namespace nerfhip {

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
    gen_ray(c2w_all, id, H, W, focal, near, far, use_ndc, ndc_plane, sx, sy, rays + r * 8);
}

}  // namespace nerfhip

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
                       nerfhip_ndc_scale(W, focal), nerfhip_ndc_scale(H, focal), near, rays_o, rays_d, out_o, out_d, n);
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
                       pixel_ids, first_pixel, n, H, W, (float)focal, near, far, use_ndc, ndc_near_plane, nerfhip_ndc_scale(W, focal),
                       nerfhip_ndc_scale(H, focal), rays, (const float*)nullptr, (float*)nullptr);
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
                       pixel_ids, (int64_t)0, n, H, W, (float)focal, near, far, use_ndc, ndc_near_plane, nerfhip_ndc_scale(W, focal),
                       nerfhip_ndc_scale(H, focal), rays, rgbs_all, rgbs);
    return nerfhip_launch_status();
}
