// Probe: does the LAYOUT of the saved-tensor blocks change the write bandwidth a chain-like kernel reaches?
// 8192 wave tiles x 160 slabs of 1 KiB (10 sections of 16 slabs = one layer's dY each), 1024 workgroups x 8 waves, ONE workgroup per CU
// (100 KB of LDS requested), every wave: for each slab { `gap` dependent FMAs; one 1 KiB store (64 lanes x 16 B, nt) }.
//   layout 0  tile-major    addr = (tile * 160 + slab) KiB                      (today: a wave fills its own 160 KiB block)
//   layout 1  section-major addr = ((sec * ntiles + tile) * 16 + slab % 16) KiB (a wave writes 16 KiB runs; neighbours adjacent)
//   layout 2  slab-major    addr = (slab * ntiles + tile) KiB
//   layout 3  section-major, workgroup-interleaved: within a section the 8 waves of a workgroup write ONE 8 KiB run per slab step
//             addr = (((sec * nwg + wg) * 16 + slab % 16) * 8 + wave) KiB
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
template <int LAYOUT>
__global__ __launch_bounds__(512) void wr(u4* __restrict__ out, long ntiles, int gap) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tile = (long)blockIdx.x * 8 + wave, nwg = ntiles / 8;
    float x = (float)lane;
    if (gap < 0) lds[threadIdx.x] = 1;      // (keeps the LDS allocation alive)
    for (int s = 0; s < 160; ++s) {
        for (int g = 0; g < gap; ++g) x = x * 1.0001f + 0.5f;
        const int sec = s >> 4, ss = s & 15;
        long piece;
        if (LAYOUT == 0) piece = tile * 160 + s;
        else if (LAYOUT == 1) piece = ((long)sec * ntiles + tile) * 16 + ss;
        else if (LAYOUT == 2) piece = (long)s * ntiles + tile;
        else piece = (((long)sec * nwg + blockIdx.x) * 16 + ss) * 8 + wave;
        u4 v = {__float_as_uint(x), (unsigned)s, (unsigned)lane, 7u};
        __builtin_nontemporal_store(v, out + piece * 64 + lane);
    }
}
int main() {
    const long ntiles = 8192;
    const size_t bytes = (size_t)160 * ntiles * 1024;
    u4* buf; if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[4] = {"tile-major", "section-major", "slab-major", "section-major, wg-interleaved"};
    for (int gap : {0, 40, 75, 110}) for (int L = 0; L < 4; ++L) {
        auto launch = [&] {
            if (L == 0) hipLaunchKernelGGL(wr<0>, dim3(ntiles / 8), dim3(512), 100 * 1024, 0, buf, ntiles, gap);
            if (L == 1) hipLaunchKernelGGL(wr<1>, dim3(ntiles / 8), dim3(512), 100 * 1024, 0, buf, ntiles, gap);
            if (L == 2) hipLaunchKernelGGL(wr<2>, dim3(ntiles / 8), dim3(512), 100 * 1024, 0, buf, ntiles, gap);
            if (L == 3) hipLaunchKernelGGL(wr<3>, dim3(ntiles / 8), dim3(512), 100 * 1024, 0, buf, ntiles, gap);
        };
        for (int r = 0; r < 3; ++r) launch();
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 8; ++r) {
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); sum += ms; if (ms < best) best = ms;
        }
        printf("gap %3d  %-32s %7.1f us avg %7.1f us min  %.2f TB/s (avg)\n", gap, names[L], sum / 8 * 1e3, best * 1e3, bytes / (sum / 8 * 1e-3) / 1e12);
    }
    return 0;
}
