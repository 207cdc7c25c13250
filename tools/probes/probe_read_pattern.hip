// Probe: does the dW kernel's HBM read rate depend on WHERE a (layer job, tile) chunk lives?  Same pipeline skeleton as
// mlp_bwd_dw_kernel<bf16> (256 workgroups x 8 waves, 4-stage LDS ring of 32 x 1 KiB pieces per stage, global_load_lds_dwordx4 nt,
// counted vmcnt + one barrier per stage, 16 MFMAs per wave per stage on the landed bytes), only the addresses differ:
//   mode 0  tile-major   (the round-1..3 layout): chunk (job, T) = 16 KiB inside tile T's 156 / 167 KiB block; two streams (dY, X)
//   mode 1  layer-major  : chunk (job, T) at (job * ntiles + T) * 16 KiB of its tensor: a workgroup's reads are two sequential streams
//   mode 2  layer-major, dY and X of a (job, T) interleaved: ONE sequential 32 KiB-per-tile stream per workgroup
// swz = lane -> 16-byte unit mapping of a piece's DMA: 0 linear, 1 the bf16 kernel's bank-conflict-free interleave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16b_nt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
constexpr int DEPTH = 4, STAGE = 32 * 1024, NJOBS = 9;
constexpr long DY_TILE = 156 * 1024, X_TILE = 167 * 1024;

template <int NMFMA>
__global__ __launch_bounds__(512, 2) void rd(const char* __restrict__ xs, const char* __restrict__ dys, long ntiles, int mode, int nsplit,
                                             int swz, float* __restrict__ out) {
    __shared__ __attribute__((aligned(1024))) char ring[DEPTH * STAGE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int job = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
    const long per = (ntiles + nsplit - 1) / nsplit, t0 = split * per;
    const long mine = t0 >= ntiles ? 0 : (ntiles - t0 < per ? ntiles - t0 : per);
    const unsigned lds = (unsigned)(uintptr_t)ring;
    const int unit = swz ? ((lane & 1) * 32 + (lane >> 1)) : lane;
    auto issue = [&](long it) {
        long T = t0 + (it < mine ? it : mine - 1);
        const unsigned slot = lds + (unsigned)((it % DEPTH) * STAGE);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pi = wave + 8 * i;                   // pieces 0..15 dY, 16..31 X
            const char* src;
            if (mode == 0) src = pi < 16 ? dys + T * DY_TILE + (long)(job * 16 + pi) * 1024 : xs + T * X_TILE + (long)(job * 16 + pi - 16) * 1024;
            else if (mode == 1) src = pi < 16 ? dys + ((long)job * ntiles + T) * 16384 + pi * 1024 : xs + ((long)job * ntiles + T) * 16384 + (pi - 16) * 1024;
            else src = xs + ((long)job * ntiles + T) * 32768 + pi * 1024;
            glds16b_nt(src + unit * 16, slot + (unsigned)(pi * 1024));
        }
    };
    f32x16 acc[4];
    for (int x = 0; x < 4; ++x) for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    for (int s = 0; s < DEPTH - 1; ++s) issue(s);
    for (long it = 0; it < mine; ++it) {
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue(it + DEPTH - 1);
        const char* st = ring + (it % DEPTH) * STAGE;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(st + wave * 2048 + lane * 16);
#pragma unroll
        for (int m = 0; m < NMFMA; ++m) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(st + 16384 + (m % 16) * 1024 + lane * 16);
            acc[m % 4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % 4], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int x = 0; x < 4; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
    if (s == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const long ntiles = 8192;
    const size_t bytes = (size_t)ntiles * X_TILE;           // >= every layout's footprint (9 jobs x 32 KiB per tile = 288 KiB in mode 2)
    char *xs, *dys; float* out;
    CK(hipMalloc(&xs, (size_t)ntiles * NJOBS * 32768)); CK(hipMalloc(&dys, bytes)); CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMemset(xs, 0, (size_t)ntiles * NJOBS * 32768)); CK(hipMemset(dys, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nsplit = 28;                                   // 9 x 28 = 252 workgroups
    const double total = (double)NJOBS * ntiles * 32768;
    for (int rep = 0; rep < 2; ++rep)
    for (int nm : {0, 16})
    for (int swz = 0; swz < 2; ++swz)
    for (int mode = 0; mode < 3; ++mode) {
        auto launch = [&] {
            if (nm == 0) hipLaunchKernelGGL(rd<0>, dim3(NJOBS * nsplit), dim3(512), 0, 0, xs, dys, ntiles, mode, nsplit, swz, out);
            else hipLaunchKernelGGL(rd<16>, dim3(NJOBS * nsplit), dim3(512), 0, 0, xs, dys, ntiles, mode, nsplit, swz, out);
        };
        for (int r = 0; r < 3; ++r) launch();
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("mfma/stage %2d  swz %d  mode %d (%s): %7.1f us  %.2f TB/s\n", nm, swz, mode,
               mode == 0 ? "tile-major " : mode == 1 ? "layer-major" : "layer-major, dY|X interleaved", ms * 1e3, total / (ms * 1e-3) / 1e12);
    }
    return 0;
}
