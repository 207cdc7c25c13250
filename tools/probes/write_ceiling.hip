// Probe: what write bandwidth does this chip sustain from compute kernels?  1.4 GB (the bytes one backward-chain launch writes):
//   fill      grid-stride 16-byte stores, plain / nt, 2048 x 256 threads
//   tile      the chain's pattern: 256 workgroups x 8 waves per round, each wave fills its own 156 KiB tile block in 1 KiB stores
//             (64 lanes x 16 B), plain / nt, with `gap` dependent FMAs between stores (0 = pure stores)
//   memset    hipMemsetAsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
template <int NT>
__global__ __launch_bounds__(256) void fill(u4* p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        u4 x = {(unsigned)i, 1u, 2u, 3u};
        if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x;
    }
}
template <int NT>
__global__ __launch_bounds__(512, 2) void tile(u4* __restrict__ out, int ns, int gap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t t = (size_t)blockIdx.x * 8 + wave;
    float x = (float)lane;
    u4* base = out + t * (size_t)ns * 64;
    for (int s = 0; s < ns; ++s) {
        for (int g = 0; g < gap; ++g) x = x * 1.0001f + 0.5f;
        u4 v = {__float_as_uint(x), (unsigned)s, (unsigned)lane, 7u};
        if (NT) __builtin_nontemporal_store(v, base + (size_t)s * 64 + lane); else base[(size_t)s * 64 + lane] = v;
    }
}
// one 16-byte store per thread, no loop (what a memset-style fill does); CONST: every thread stores the same value
template <int CONST_>
__global__ __launch_bounds__(256) void fill1(u4* p, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { u4 x = CONST_ ? u4{1u, 1u, 1u, 1u} : u4{(unsigned)i, 1u, 2u, 3u}; p[i] = x; }
}
// each thread stores 4 consecutive 16-byte units (64 B per lane, 4 KiB per wave contiguous)
__global__ __launch_bounds__(256) void fill4(u4* p, size_t n4) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p[i + k] = u4{(unsigned)i, 1u, 2u, (unsigned)k};
    }
}
// the tile pattern through raw buffer stores with the cache-policy bits given (what the kernels use: aux 2 = nt)
template <int AUX>
__global__ __launch_bounds__(512, 2) void tile_buf(u4* __restrict__ out, int ns) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t t = (size_t)blockIdx.x * 8 + wave;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + t * (size_t)ns * 64, 0, ns * 1024, 0x00020000);
    for (int s = 0; s < ns; ++s) {
        u4 v = {(unsigned)t, (unsigned)s, (unsigned)lane, 7u};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)(s * 1024 + lane * 16), 0, AUX);
    }
}
int main() {
    const int ns = 156; const size_t ntiles = 8192;                      // 8192 tiles x 156 KiB = 1.31 GB
    const size_t bytes = (size_t)ns * ntiles * 1024;
    u4* buf; if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        for (int r = 0; r < 3; ++r) launch();
        float best = 1e9f, sum = 0.f;
        for (int r = 0; r < 8; ++r) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); sum += ms; if (ms < best) best = ms;
        }
        printf("%-44s %8.1f us avg %8.1f us min  %.2f TB/s (min)\n", name, sum / 8 * 1e3, best * 1e3, bytes / (best * 1e-3) / 1e12);
    };
    timeit("fill plain   (2048 x 256 threads)", [&] { hipLaunchKernelGGL(fill<0>, dim3(2048), dim3(256), 0, 0, buf, bytes / 16); });
    timeit("fill nt      (2048 x 256 threads)", [&] { hipLaunchKernelGGL(fill<1>, dim3(2048), dim3(256), 0, 0, buf, bytes / 16); });
    timeit("fill plain   (8192 x 256 threads)", [&] { hipLaunchKernelGGL(fill<0>, dim3(8192), dim3(256), 0, 0, buf, bytes / 16); });
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(buf, 1, bytes, 0); });
    timeit("fill1 one store per thread", [&] { hipLaunchKernelGGL(fill1<0>, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, 0, buf, bytes / 16); });
    timeit("fill1 one store per thread, constant", [&] { hipLaunchKernelGGL(fill1<1>, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, 0, buf, bytes / 16); });
    timeit("fill4 64 B per lane", [&] { hipLaunchKernelGGL(fill4, dim3((unsigned)(bytes / 64 / 256)), dim3(256), 0, 0, buf, bytes / 16); });
    timeit("tile buffer_store aux 0", [&] { hipLaunchKernelGGL(tile_buf<0>, dim3(ntiles / 8), dim3(512), 0, 0, buf, ns); });
    timeit("tile buffer_store aux 1 (sc0)", [&] { hipLaunchKernelGGL(tile_buf<1>, dim3(ntiles / 8), dim3(512), 0, 0, buf, ns); });
    timeit("tile buffer_store aux 2 (nt)", [&] { hipLaunchKernelGGL(tile_buf<2>, dim3(ntiles / 8), dim3(512), 0, 0, buf, ns); });
    timeit("tile buffer_store aux 3 (sc0 nt)", [&] { hipLaunchKernelGGL(tile_buf<3>, dim3(ntiles / 8), dim3(512), 0, 0, buf, ns); });
    timeit("tile buffer_store aux 16 (sc1)", [&] { hipLaunchKernelGGL(tile_buf<16>, dim3(ntiles / 8), dim3(512), 0, 0, buf, ns); });
    timeit("tile buffer_store aux 18 (sc1 nt)", [&] { hipLaunchKernelGGL(tile_buf<18>, dim3(ntiles / 8), dim3(512), 0, 0, buf, ns); });
    for (int gap : {0, 8, 30}) {
        char nm[96];
        snprintf(nm, sizeof nm, "tile plain gap %d (1024 wgs x 8 waves)", gap);
        timeit(nm, [&] { hipLaunchKernelGGL(tile<0>, dim3(ntiles / 8), dim3(512), 0, 0, buf, ns, gap); });
        snprintf(nm, sizeof nm, "tile nt    gap %d (1024 wgs x 8 waves)", gap);
        timeit(nm, [&] { hipLaunchKernelGGL(tile<1>, dim3(ntiles / 8), dim3(512), 0, 0, buf, ns, gap); });
    }
    return 0;
}
