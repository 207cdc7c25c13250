// What can a register-resident MFMA chain of the forward kernel's shape sustain on a whole MI355X?
// Each wave runs ITER x 16 v_mfma_f32_32x32x16_bf16 in one of several instruction mixes; the host reports chip-wide
// TFLOP/s, and the kernel reports the shader clock it actually ran at (clock64 = s_memtime vs wall_clock64 = 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o probe_mfma_rate tools/probes/probe_mfma_rate.hip && ./probe_mfma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kLds = 152 * 1024;      // one workgroup per CU, as the real kernel

// MODE bits: 1 = alternate the two accumulators every MFMA (else: 16 chained MFMAs per accumulator, then the other — the
//            forward kernel's tile-major order), 2 = one ds_read_b128 (A fragment) per MFMA, 4 = one per TWO MFMAs (each
//            fragment feeds two accumulators: the 64-points-per-wave shape), 8 = 2 VALU per MFMA on the idle accumulator,
//            16 = the 16 B operands differ (else one)
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void k(float* out, uint64_t* clk, int iters) {
    __shared__ __attribute__((aligned(1024))) char lds[kLds];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += NW * 64) reinterpret_cast<float*>(lds)[i] = 1e-3f * (i & 255);
    __syncthreads();
    const char* base = lds + lane * 16;
    bf16x8 b[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(0.01f * (lane + j + e));
    f32x16 acc0 = {}, acc1 = {};
    bf16x8 a[2];
    unsigned sink = 0;
    a[0] = *reinterpret_cast<const bf16x8*>(base);
    a[1] = *reinterpret_cast<const bf16x8*>(base + 1024);
    const uint64_t c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const bool first = (MODE & 1) ? (i & 1) == 0 : (MODE & 4) ? (i & 1) == 0 : i < 16;
            const bf16x8 bb = b[(MODE & 16) ? ((MODE & 4) ? (i >> 1) & 15 : i & 15) : 0];
            const int sl = (MODE & 4) ? (i >> 1) & 1 : i & 1;
            if (first) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl], bb, acc0, 0, 0, 0);
            else       acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl], bb, acc1, 0, 0, 0);
            if ((MODE & 2) || ((MODE & 4) && (i & 1)))
                a[sl] = *reinterpret_cast<const bf16x8*>(base + ((it & 2) ? 32768 : 0) + ((i + 2) & 31) * 1024);
            if (MODE & 8) {
                const f32x16& o = first ? acc1 : acc0;
                const f32x2 xv = {o[i & 15], o[(i + 1) & 15]};
                bf16x2 pk = __builtin_convertvector(xv, bf16x2);
                s16x2 sv = __builtin_bit_cast(s16x2, pk);
                const s16x2 z = {0, 0};
                sv = __builtin_elementwise_max(sv, z);
                sink ^= __builtin_bit_cast(unsigned, sv);
            }
            if (MODE & 14) __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s + (float)sink;
    if ((threadIdx.x & 63) == 0) {      // per wave: cycles, wall ticks, start offset
        const int w = blockIdx.x * NW + (threadIdx.x >> 6);
        clk[2 * w] = c1 - c0; clk[2 * w + 1] = w1 - w0;
    }
}

template <int MODE, int NW>
void run(const char* name, int blocks, int iters) {
    float* out; uint64_t* clk;
    hipMalloc(&out, (size_t)blocks * NW * 64 * 4);
    hipMalloc(&clk, (size_t)blocks * NW * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, NW>), dim3(blocks), dim3(NW * 64), 0, 0, out, clk, iters);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE, NW>), dim3(blocks), dim3(NW * 64), 0, 0, out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * blocks * NW);
    hipMemcpy(h.data(), clk, (size_t)blocks * NW * 16, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < blocks * NW; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
    const double us = ms * 1e3 / reps;
    const double flop = (double)blocks * NW * iters * 16 * 32768.0;
    const double mhz = cyc / wall * 100.0;
    const double cyc_per_mfma = (cyc / blocks / NW) / ((double)iters * 16);   // per wave, start to end
    printf("%-58s NW=%d blocks=%d: %8.1f us  %7.1f TFLOP/s (%4.1f %% of 2500)  clock %6.0f MHz  %5.1f wave-cyc/MFMA\n", name, NW, blocks,
           us, flop / us * 1e-6, flop / us * 1e-6 / 25.0, mhz, cyc_per_mfma);
    hipFree(out); hipFree(clk);
}

int main() {
    const int it = 74;      // 1184 MFMAs per wave = one pass of the NeRF MLP
    const int blocks = 768;
    run<0, 4>("tile-major chain (16 per accumulator), mfma only", blocks, it);
    run<0, 8>("tile-major chain (16 per accumulator), mfma only", blocks, it);
    run<1, 4>("alternating accumulators, mfma only", blocks, it);
    run<1, 8>("alternating accumulators, mfma only", blocks, it);
    run<16, 8>("tile-major, 16 B operands", blocks, it);
    run<2 + 16, 8>("tile-major + ds_read_b128 per mfma", blocks, it);
    run<2 + 8 + 16, 8>("tile-major + ds_read_b128 + 2 VALU per mfma  [= the pipelined forward]", blocks, it);
    run<1 + 2 + 8 + 16, 8>("alternating + ds_read_b128 + 2 VALU per mfma", blocks, it);
    run<4 + 16, 4>("A fragment feeds two accumulators (64 points / wave): ds_read per 2 mfma", blocks, it);
    run<4 + 8 + 16, 4>("  + 2 VALU per mfma", blocks, it);
    run<4 + 8 + 16, 8>("  + 2 VALU per mfma, two such waves per SIMD", blocks, it);
    run<2 + 8 + 16, 8>("long run (10 passes): tile-major + ds_read + 2 VALU", blocks, 740);
    run<4 + 8 + 16, 4>("long run (10 passes): 64 points / wave", blocks, 740);
    return 0;
}
