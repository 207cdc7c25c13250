// Probe: HBM write bandwidth of the activation-store pattern.  Each wave writes NS pieces of 1 KiB (64 lanes x 16 B).
//   mode 0: tile-major   addr = (tile * NS + s) * 1 KiB   (what mlp_fwd SAVE / mlp_bwd_chain do: a wave fills its own block)
//   mode 1: slab-major   addr = (s * ntiles + tile) * 1 KiB (all waves write slab s of their tiles next to each other)
// `gap` = dependent VALU iterations between stores (emulates the MFMA work between epilogues).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512, 2) void wr(uint4* __restrict__ out, int ns, long ntiles, int mode, int gap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tile = (long)blockIdx.x * 8 + wave;
    float x = (float)lane;
    for (int s = 0; s < ns; ++s) {
        for (int g = 0; g < gap; ++g) x = x * 1.0001f + 0.5f;
        const long piece = mode == 0 ? tile * ns + s : (long)s * ntiles + tile;
        uint4 v = make_uint4(__float_as_uint(x), s, lane, 7);
        out[piece * 64 + lane] = v;
    }
}
int main(int argc, char** argv) {
    const int ns = 167; const long ntiles = 6144;
    uint4* buf; hipMalloc(&buf, (size_t)ns * ntiles * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int gap : {0, 200, 1000}) for (int mode = 0; mode < 2; ++mode) {
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(wr, dim3(ntiles / 8), dim3(512), 0, 0, buf, ns, ntiles, mode, gap);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(wr, dim3(ntiles / 8), dim3(512), 0, 0, buf, ns, ntiles, mode, gap);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("gap %4d mode %d (%s): %.1f us  %.2f TB/s\n", gap, mode, mode ? "slab-major" : "tile-major", ms * 1e3,
               (double)ns * ntiles * 1024 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
