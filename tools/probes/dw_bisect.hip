// Bisect from the dW kernel's pipeline skeleton (tools/probes/probe_read_pattern.hip: 6.7-6.9 TB/s) towards the product kernel
// (mlp_bwd_dw_kernel<bf16>: 5.3-5.9 TB/s), ONE feature at a time.  Every variant streams the same bytes
// — 9 jobs x 8192 tiles x 32 KiB through 252 workgroups of 8 waves, tile-major addresses, LDS-DMA nt, counted vmcnt + one barrier
// per stage — and differs in what the waves do with a landed stage:
//   DATA    0 zeros (the skeleton's buffers) | 1 random bf16 in [-1, 1) (the MFMA and HBM buses toggle: the power the product pays)
//   READ    0 ds_read_b128 lane-linear fragments | 1 the product's two ds_read_b64_tr_b16 per fragment on its swizzled piece image
//   ACC     0 four accumulator tiles, m % 4 | 1 the product's mapping: 8 X tiles x 2 k-steps, A = the wave's own dY tile
//   PIPE    0 compiler-scheduled | 1 the product's pinned software pipeline (RD fragments in flight, sched_barrier per MFMA)
//   SPREAD  0 the next stage's DMAs in one block after the barrier | 1 one DMA every 4 MFMAs
//   DEPTH   ring stages (4 = skeleton, 5 = product)
//   SPLIT   0 wave = 1 dY tile x 8 X tiles (18 KiB of LDS reads per wave per stage) | 1 wave = 2 dY tiles x 4 X tiles (12 KiB)
//   BIAS    1 = the 16 VALU bias sums per stage (serial cvt + add chain) | 2 = four v_dot2_f32_bf16 against (1, 1) per k-step, two chains
//           | 3 = one more MFMA per k-step against an all-ones B fragment (no VALU at all)
//   EPI     1 = the partial-slab epilogue (128 KiB of fp32 per workgroup; lane-major: a lane's 16 floats contiguous, the product's layout)
//           | 2 = the same bytes, register-major (every store instruction writes 1 KiB contiguous) | 3 = register-major, nt
//           | 4 = one float per wave
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dw_bisect tools/probes/dw_bisect.hip ; run on the GPU box: ./dw_bisect
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16b_nt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
constexpr int STAGE = 32 * 1024, NJOBS = 9;
constexpr long DY_TILE = 156 * 1024, X_TILE = 167 * 1024;

template <int N>
__device__ __forceinline__ void wait_vm_barrier() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); }

template <int READ, int ACC, int PIPE, int SPREAD, int DEPTH, int SPLIT, int BIAS, int EPI>
__global__ __launch_bounds__(512, 2) void rd(const char* __restrict__ xs, const char* __restrict__ dys, long ntiles, int nsplit,
                                             float* __restrict__ out) {
    __shared__ __attribute__((aligned(1024))) char ring[DEPTH * STAGE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int job = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
    const long per = (ntiles + nsplit - 1) / nsplit, t0 = split * per;
    const long mine = t0 >= ntiles ? 0 : (ntiles - t0 < per ? ntiles - t0 : per);
    const unsigned lds = (unsigned)(uintptr_t)ring;
    // lane -> 16-byte unit of a piece: READ 0 linear; READ 1 the product's image (unit (2n+h) for even slabs, ^8 for odd ones)
    const int dma_even = READ ? ((lane & 1) * 32 + (lane >> 1)) * 16 : lane * 16;
    const int dma_odd = READ ? ((lane & 1) * 32 + ((lane ^ 8) >> 1)) * 16 : lane * 16;
    int s_issue = 0, s_use = 0;
    const char* dbase = nullptr;
    const char* abase = nullptr;
    unsigned slot = 0;
    auto next_stage = [&](long it) {
        long T = t0 + (it < mine ? it : mine - 1);
        dbase = dys + T * DY_TILE + (long)job * 16 * 1024;
        abase = xs + T * X_TILE + (long)job * 16 * 1024;
        slot = lds + (unsigned)(s_issue * STAGE);
        s_issue = (s_issue + 1 == DEPTH) ? 0 : s_issue + 1;
    };
    auto issue_piece = [&](int i) {
        const int pi = wave + 8 * i;                       // pieces 0..15 dY, 16..31 X
        const char* src = pi < 16 ? dbase + pi * 1024 : abase + (pi - 16) * 1024;
        glds16b_nt(src + ((pi & 1) ? dma_odd : dma_even), slot + (unsigned)(pi * 1024));
    };
    auto issue_stage = [&](long it) {
        next_stage(it);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(i);
    };
    const int grp = lane >> 4, c = lane & 15;
    const int tr_off = (grp & 1) * 1024 + (2 * (8 * (grp >> 1) + (c >> 2)) + ((c & 3) & 1)) * 16 + ((c & 3) >> 1) * 8;
    const int tr_s0 = (grp & 1) ? 128 : 0, tr_s1 = 128 - tr_s0;
    auto load_frag = [&](const char* pb, int q) -> bf16x8 {          // pb = a 2 KiB slab pair (32 features x 32 points), q = k-step
        if constexpr (READ) {
            union { s16x4 h2[2]; bf16x8 v; } f;
            f.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + tr_off + q * 512 + tr_s0));
            f.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + tr_off + q * 512 + tr_s1));
            return f.v;
        } else {
            return *reinterpret_cast<const bf16x8*>(pb + q * 1024 + lane * 16);
        }
    };
    constexpr int NACC = ACC ? 8 : 4;
    f32x16 acc[NACC];
    for (int x = 0; x < NACC; ++x) for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    float dbacc = 0.f;
    [[maybe_unused]] float dbacc2 = 0.f;
    [[maybe_unused]] f32x16 acc_b;
    for (int r = 0; r < 16; ++r) acc_b[r] = 0.f;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x8 ones;
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    auto bias_sum = [&](const bf16x8& a) {
        if constexpr (BIAS == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dbacc += (float)a[j];
        } else if constexpr (BIAS == 2) {
            const bf16x2 o = {(__bf16)1.0f, (__bf16)1.0f};
            dbacc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{a[0], a[1]}, o, dbacc, false);
            dbacc2 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{a[2], a[3]}, o, dbacc2, false);
            dbacc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{a[4], a[5]}, o, dbacc, false);
            dbacc2 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{a[6], a[7]}, o, dbacc2, false);
        } else if constexpr (BIAS == 3) {
            acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, acc_b, 0, 0, 0);
        }
    };
    for (int s = 0; s < DEPTH - 1; ++s) issue_stage(s);
    for (long it = 0; it < mine; ++it) {
        wait_vm_barrier<(DEPTH - 2) * 4>();
        if (SPREAD) next_stage(it + DEPTH - 1); else issue_stage(it + DEPTH - 1);
        const char* st = ring + s_use * STAGE;
        s_use = (s_use + 1 == DEPTH) ? 0 : s_use + 1;
        const char* x_base = st + 16 * 1024;
        if constexpr (!ACC) {
            const bf16x8 a = load_frag(st + wave * 2048, 0);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const bf16x8 b = load_frag(x_base + (m % 8) * 2048, m / 8);
                acc[m % 4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % 4], 0, 0, 0);
                if (SPREAD && m % 4 == 3) issue_piece(m / 4);
            }
        } else if constexpr (SPLIT) {
            // wave (i = wave >> 1, j = wave & 1): dY tiles 2i, 2i+1 against X tiles 4j .. 4j+3: acc[2 * xt + dt]
            const char* dy_base = st + (wave >> 1) * 4096;
            const char* xb = x_base + (wave & 1) * 8192;
            if constexpr (PIPE) {
                constexpr int RD = 4;                                  // B fragments in flight
                bf16x8 a[2][2], b[RD];
                a[0][0] = load_frag(dy_base, 0);
                a[0][1] = load_frag(dy_base + 2048, 0);
#pragma unroll
                for (int f = 0; f < RD - 1; ++f) b[f] = load_frag(xb + (f % 4) * 2048, f / 4);
                a[1][0] = load_frag(dy_base, 1);
                a[1][1] = load_frag(dy_base + 2048, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < 8; ++f) {                          // B fragment f = (k-step f / 4, X tile f % 4): 2 MFMAs each
                    if (f + RD - 1 < 8) b[(f + RD - 1) % RD] = load_frag(xb + ((f + RD - 1) % 4) * 2048, (f + RD - 1) / 4);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[2 * (f % 4)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[f / 4][0], b[f % RD], acc[2 * (f % 4)], 0, 0, 0);
                    acc[2 * (f % 4) + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[f / 4][1], b[f % RD], acc[2 * (f % 4) + 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (BIAS && (f == 1 || f == 5)) bias_sum(a[f / 4][0]);
                    if (SPREAD && f % 2 == 1) { issue_piece(f / 2); __builtin_amdgcn_sched_barrier(0); }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const bf16x8 a0 = load_frag(dy_base, q), a1 = load_frag(dy_base + 2048, q);
                    if (BIAS) bias_sum(a0);
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const bf16x8 b = load_frag(xb + x * 2048, q);
                        acc[2 * x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[2 * x], 0, 0, 0);
                        acc[2 * x + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[2 * x + 1], 0, 0, 0);
                        if (SPREAD && x % 2 == 1) issue_piece(q * 2 + x / 2);
                    }
                }
            }
        } else {
            const char* dy_base = st + wave * 2048;
            if constexpr (PIPE) {
                constexpr int RD = 5, NM = 16, NXT = 8;
                const bf16x8 a0 = load_frag(dy_base, 0);
                bf16x8 b[RD];
#pragma unroll
                for (int m = 0; m < RD - 1; ++m) b[m] = load_frag(x_base + (m % NXT) * 2048, m / NXT);
                const bf16x8 a1 = load_frag(dy_base, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int x = m % NXT;
                    if (m + RD - 1 < NM) b[(m + RD - 1) % RD] = load_frag(x_base + ((m + RD - 1) % NXT) * 2048, (m + RD - 1) / NXT);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m < NXT ? a0 : a1, b[m % RD], acc[x], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (BIAS && m == 1) { bias_sum(a0); if (BIAS == 3) __builtin_amdgcn_sched_barrier(0); }
                    if (BIAS && m == NXT + 1) { bias_sum(a1); if (BIAS == 3) __builtin_amdgcn_sched_barrier(0); }
                    if (SPREAD && m % 4 == 3) { issue_piece(m / 4); __builtin_amdgcn_sched_barrier(0); }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const bf16x8 a = load_frag(dy_base, q);
                    if (BIAS) bias_sum(a);
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        const bf16x8 b = load_frag(x_base + x * 2048, q);
                        acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[x], 0, 0, 0);
                        if (SPREAD && x % 4 == 3) issue_piece(q * 2 + x / 4);
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (BIAS == 2) dbacc += dbacc2;
    if (BIAS == 3) dbacc = acc_b[0];
    if constexpr (EPI == 1) {
        float* sl = out + (size_t)blockIdx.x * (8 * NACC * 64 * 16 + 512);
#pragma unroll
        for (int x = 0; x < NACC; ++x) {
            float* dst = sl + ((size_t)(wave * NACC + x) * 64 + lane) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                reinterpret_cast<float4*>(dst)[q] = make_float4(acc[x][4 * q], acc[x][4 * q + 1], acc[x][4 * q + 2], acc[x][4 * q + 3]);
        }
        sl[8 * NACC * 64 * 16 + wave * 64 + lane] = dbacc;
    } else if constexpr (EPI == 2 || EPI == 3) {
        float* sl = out + (size_t)blockIdx.x * (8 * NACC * 64 * 16 + 512);
#pragma unroll
        for (int x = 0; x < NACC; ++x) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                typedef __attribute__((ext_vector_type(4))) float f32x4;
                f32x4* dst = reinterpret_cast<f32x4*>(sl + ((size_t)((wave * NACC + x) * 4 + q) * 64 + lane) * 4);
                const f32x4 v = {acc[x][4 * q], acc[x][4 * q + 1], acc[x][4 * q + 2], acc[x][4 * q + 3]};
                if (EPI == 3) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        }
        sl[8 * NACC * 64 * 16 + wave * 64 + lane] = dbacc;
    } else if constexpr (EPI == 4) {
        float s = dbacc;
        for (int x = 0; x < NACC; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
        if (lane == 0) out[blockIdx.x * 8 + wave] = s;
    } else {
        float s = dbacc;
        for (int x = 0; x < NACC; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
        if (s == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

__global__ void fill_random(uint32_t* p, size_t n_dwords, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_dwords; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        // two bf16 in [-1, 1): sign + exponent 0x3f0..0x3f7 region -> keep exponents in [2^-8, 1)
        const uint32_t lo = (h & 0x807f) | ((0x77 + ((h >> 8) & 7)) << 7);
        const uint32_t hi = ((h >> 16) & 0x807f) | ((0x77 + ((h >> 24) & 7)) << 7);
        p[i] = lo | (hi << 16);
    }
}

struct Variant { const char* name; void (*launch)(const char*, const char*, long, int, float*); };
template <int READ, int ACC, int PIPE, int SPREAD, int DEPTH, int SPLIT, int BIAS, int EPI>
void launch_v(const char* xs, const char* dys, long ntiles, int nsplit, float* out) {
    hipLaunchKernelGGL((rd<READ, ACC, PIPE, SPREAD, DEPTH, SPLIT, BIAS, EPI>), dim3(NJOBS * nsplit), dim3(512), 0, 0, xs, dys, ntiles, nsplit, out);
}

int main(int argc, char** argv) {
    const long ntiles = 8192;
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    char *xs[2], *dys[2]; float* out;
    for (int d = 0; d < 2; ++d) {
        CK(hipMalloc(&xs[d], (size_t)ntiles * X_TILE)); CK(hipMalloc(&dys[d], (size_t)ntiles * DY_TILE));
        if (d == 0) { CK(hipMemset(xs[d], 0, (size_t)ntiles * X_TILE)); CK(hipMemset(dys[d], 0, (size_t)ntiles * DY_TILE)); }
        else {
            hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (uint32_t*)xs[d], (size_t)ntiles * X_TILE / 4, 1u);
            hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (uint32_t*)dys[d], (size_t)ntiles * DY_TILE / 4, 2u);
        }
    }
    CK(hipMalloc(&out, (size_t)256 * (8 * 8 * 64 * 16 + 512) * 4));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nsplit = 28;                                   // 9 x 28 = 252 workgroups
    const double total = (double)NJOBS * ntiles * 32768;
    //                                             READ ACC PIPE SPREAD DEPTH SPLIT BIAS EPI
    const Variant vs[] = {
        {"0 skeleton: b128 reads, 4 acc, depth 4          ", launch_v<0, 0, 0, 0, 4, 0, 0, 0>},
        {"1 + tr16_b64 reads on the swizzled image       ", launch_v<1, 0, 0, 0, 4, 0, 0, 0>},
        {"2 + product MFMA mapping (8 X tiles x 2 k-steps)", launch_v<1, 1, 0, 0, 4, 0, 0, 0>},
        {"3 + pinned pipeline RD 5                        ", launch_v<1, 1, 1, 0, 4, 0, 0, 0>},
        {"4 + DMAs spread between the MFMAs               ", launch_v<1, 1, 1, 1, 4, 0, 0, 0>},
        {"5 + ring depth 5                                ", launch_v<1, 1, 1, 1, 5, 0, 0, 0>},
        {"6 + bias sums                                   ", launch_v<1, 1, 1, 1, 5, 0, 1, 0>},
        {"7 + partial-slab epilogue  (= product shape)    ", launch_v<1, 1, 1, 1, 5, 0, 1, 1>},
        {"8 product shape, compiler-scheduled             ", launch_v<1, 1, 0, 1, 5, 0, 1, 1>},
        {"9 product shape, DMAs in one block              ", launch_v<1, 1, 1, 0, 5, 0, 1, 1>},
        {"F product shape, epilogue register-major        ", launch_v<1, 1, 1, 1, 5, 0, 1, 2>},
        {"G product shape, epilogue register-major nt     ", launch_v<1, 1, 1, 1, 5, 0, 1, 3>},
        {"H product shape, epilogue = one float per wave  ", launch_v<1, 1, 1, 1, 5, 0, 1, 4>},
        {"I product shape, bias by v_dot2, lane-major epi ", launch_v<1, 1, 1, 1, 5, 0, 2, 1>},
        {"J product shape, bias by a ones MFMA            ", launch_v<1, 1, 1, 1, 5, 0, 3, 1>},
        {"K product shape, dot2 bias + register-major epi ", launch_v<1, 1, 1, 1, 5, 0, 2, 2>},
        {"L product shape, MFMA bias + register-major epi ", launch_v<1, 1, 1, 1, 5, 0, 3, 2>},
        {"M 2x4 split, dot2 bias + register-major epi     ", launch_v<1, 1, 1, 1, 5, 1, 2, 2>},
        {"A 2x4 wave split, pinned, spread, depth 5       ", launch_v<1, 1, 1, 1, 5, 1, 1, 1>},
        {"B 2x4 wave split, compiler-scheduled, depth 5   ", launch_v<1, 1, 0, 1, 5, 1, 1, 1>},
        {"C 2x4 wave split, pinned, block DMAs, depth 5   ", launch_v<1, 1, 1, 0, 5, 1, 1, 1>},
        {"D 2x4 wave split, b128 reads (no transpose)     ", launch_v<0, 1, 0, 1, 5, 1, 1, 1>},
        {"E 1x8 mapping, b128 reads (no transpose)        ", launch_v<0, 1, 0, 1, 5, 0, 1, 1>},
    };
    for (int rep = 0; rep < 2; ++rep)
        for (int data = 0; data < 2; ++data)
            for (const Variant& v : vs) {
                for (int r = 0; r < 5; ++r) v.launch(xs[data], dys[data], ntiles, nsplit, out);
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; ++r) v.launch(xs[data], dys[data], ntiles, nsplit, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
                printf("rep %d data %s  %s: %7.1f us  %.2f TB/s\n", rep, data ? "random" : "zeros ", v.name, ms * 1e3, total / (ms * 1e-3) / 1e12);
                fflush(stdout);
            }
    return 0;
}
