// nn.Linear (+ ReLU / Sigmoid) forward and backward, one MFMA GEMM launch per layer: the path of a NeRF whose shape is NOT the
// reference's default (models/nerf.py:42-81 takes any D, W, skips, channel counts; the fused register-resident kernels of
// mlp_fwd_kernel.h / mlp_bwd*.hip are built for D=8, W=256, skips=[4], 63/27 only).  Activations round-trip HBM between layers
// here, exactly as in the reference's cuBLAS path — this is API completeness for non-default shapes, not the
// benchmarked path.
//
// One kernel template,  C[i][j] = sum_k A(i, k) * B(j, k),  128 x 128 or 128 x 256 output tile per workgroup (4 / 8 waves, a 64 x 64
// sub-tile of 2 x 2 MFMA blocks each), 32 reduction steps per LDS stage (fp32 -> bf16 conversion on the way into LDS, 16-byte global loads
// where alignment allows), each operand either k-contiguous or row-contiguous in HBM, so that the three GEMMs of a layer are
// the same code:
//     forward        y[m][f]   = act( x[m][:] . W[f][:] (+ y[m][f]) + b[f] )        i = point,   j = out feature, k = in feature
//     input grad     gx[m][c]  = sum_f g[m][f] W[f][c]                              i = point,   j = in feature,  k = out feature
//     weight grad    gW[f][c]  = sum_m g[m][f] x[m][c];  gb[f] = sum_m g[m][f]      i = out feat, j = in feature, k = point (split
//                                                                                    over k; gb = row sums of the A fragments)
// with g = gy * act'(y) formed while the operand is loaded (ReLU: y > 0; Sigmoid: y (1 - y), torch's sigmoid_backward).
// Precision: NERFHIP_F32 = v_mfma_f32_32x32x2_f32 (every product and sum in fp32); NERFHIP_BF16 / _BF16_F8 = operands rounded to
// bf16 (RNE) on their way into LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulation.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace nerfhip {
namespace lin {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kTileI = 128;  // output tile: 128 rows (i) x 128 or 256 columns (j); every wave a 64 x 64 sub-tile (2 x 2 MFMA blocks)
constexpr int kStage = 32;   // reduction steps per LDS stage
#ifndef NERFHIP_LIN_DEPTH
#define NERFHIP_LIN_DEPTH 1  // 2 (a second register buffer) costs the second wave per SIMD: 192-248 VGPRs + 64 accumulators
#endif
constexpr int kDepth = NERFHIP_LIN_DEPTH;

// One GEMM operand: a (rows x K) matrix of which the kernel needs tiles of 128 rows x 32 reduction steps.
//   ROWC = false ("k-contiguous"):   element (r, k) = p[r * s + k]      activations x / g as the point-major side
//   ROWC = true  ("row-contiguous"): element (r, k) = p[r + k * s]      the same tensors when the POINTS are the reduction index,
//                                                                       and W when the output features are
// `y` (same addressing with stride ys): the layer OUTPUT belonging to a gradient operand; the element is multiplied by act'(y).
// vec: pointer(s) 16-byte aligned and stride(s) a multiple of 4 floats => 16-byte loads (a float4 that starts inside the matrix
// ends inside the same row of the underlying tensor, because that row is a whole number of float4s).
struct Operand {
    const float* p;
    int64_t s;
    const float* y;
    int64_t ys;
    int act;
    int64_t rows;
    int vec;
};

__device__ __forceinline__ float act_grad(float g, float y, int act) {
    if (act == NERFHIP_ACT_RELU) return y > 0.0f ? g : 0.0f;
    if (act == NERFHIP_ACT_SIGMOID) return nh_mul(nh_mul(g, nh_sub(1.0f, y)), y);
    return g;
}

// LDS image of one operand tile of ROWS rows.  k-contiguous: [row][k], pitch 40 bf16 (80 B) / 33 fp32 — one ds_read_b128 (bf16) or
// conflict-free ds_read_b32 (fp32) per fragment.  Row-contiguous: [k][row], pitch ROWS + 32 bf16 (320 / 576 B = 16 banks mod 64: the
// 4 k-rows x 2 feature halves of a 32-lane ds_read_b64_tr_b16 group land on 32 distinct bank pairs) / ROWS fp32.
template <bool F32, bool ROWC, int ROWS> struct Image {
    using T = typename std::conditional<F32, float, __bf16>::type;
    static constexpr int pitch = ROWC ? (F32 ? ROWS : ROWS + 32) : (F32 ? 33 : 40);
    static constexpr int elems = ROWC ? kStage * pitch : ROWS * pitch;
};

// The tile (ROWS rows x 32 k) is ROWS * 8 units of 4 elements, UPT per thread.  Unit e of thread t:
//   k-contiguous:   row e * NT/8 + t/8,           k 4 (t % 8) .. +3        (8 lanes read 128 contiguous bytes of a row)
//   row-contiguous: rows 4 (t % (ROWS/4)) .. +3,  k e * 4 NT/ROWS + t / (ROWS/4)
template <bool ROWC, int ROWS, int NT> struct Units {
    static constexpr int UPT = ROWS * 8 / NT;
    static constexpr int UPK = ROWS / 4;
    __device__ static __forceinline__ void coord(int e, int t, int& r, int& k) {
        if (ROWC) {
            r = (t % UPK) * 4;
            k = e * (NT / UPK) + t / UPK;
        } else {
            r = e * (NT / 8) + (t >> 3);
            k = (t & 7) * 4;
        }
    }
};

template <bool ROWC, int ROWS, int NT, bool VEC>
__device__ __forceinline__ void fetch(const Operand& O, int64_t r0, int64_t k0, int64_t kend, float (&v)[Units<ROWC, ROWS, NT>::UPT][4]) {
    using U = Units<ROWC, ROWS, NT>;
    const int t = (int)threadIdx.x;
    const float* base = ROWC ? O.p + r0 + k0 * O.s : O.p + r0 * O.s + k0;
    const float* ybase = O.y ? (ROWC ? O.y + r0 + k0 * O.ys : O.y + r0 * O.ys + k0) : nullptr;
    const int s32 = (int)O.s, ys32 = (int)O.ys;
#pragma unroll
    for (int e = 0; e < U::UPT; ++e) {
        int r, k;
        U::coord(e, t, r, k);
        const int off = ROWC ? r + k * s32 : r * s32 + k;
        const int yoff = ROWC ? r + k * ys32 : r * ys32 + k;
        // the unit runs along rows (ROWC) or along k: `lim` = its number of valid elements (may be <= 0 or > 4)
        const int64_t lim = ROWC ? O.rows - (r0 + r) : kend - (k0 + k);
        const bool ok = ROWC ? (k0 + k < kend) : (r0 + r < O.rows);
        float x[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (ok && lim > 0) {
            if (VEC || O.vec) {
                const float4 q = *reinterpret_cast<const float4*>(base + off);
                x[0] = q.x, x[1] = q.y, x[2] = q.z, x[3] = q.w;
                if (ybase) {
                    const float4 yq = *reinterpret_cast<const float4*>(ybase + yoff);
                    x[0] = act_grad(x[0], yq.x, O.act), x[1] = act_grad(x[1], yq.y, O.act);
                    x[2] = act_grad(x[2], yq.z, O.act), x[3] = act_grad(x[3], yq.w, O.act);
                }
#pragma unroll
                for (int q4 = 1; q4 < 4; ++q4)
                    if (q4 >= lim) x[q4] = 0.0f;
            } else {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    if (q4 < lim) {
                        x[q4] = base[off + q4];
                        if (ybase) x[q4] = act_grad(x[q4], ybase[yoff + q4], O.act);
                    }
            }
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) v[e][q4] = x[q4];
    }
}

template <bool F32, bool ROWC, int ROWS, int NT>
__device__ __forceinline__ void stash(typename Image<F32, ROWC, ROWS>::T* S, const float (&v)[Units<ROWC, ROWS, NT>::UPT][4]) {
    using U = Units<ROWC, ROWS, NT>;
    constexpr int P = Image<F32, ROWC, ROWS>::pitch;
    const int t = (int)threadIdx.x;
#pragma unroll
    for (int e = 0; e < U::UPT; ++e) {
        int r, k;
        U::coord(e, t, r, k);
        const int idx = ROWC ? k * P + r : r * P + k;
        if constexpr (F32) {
            if constexpr (ROWC) {
                *reinterpret_cast<float4*>(S + idx) = make_float4(v[e][0], v[e][1], v[e][2], v[e][3]);   // pitch ROWS: 16-B aligned
            } else {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) S[idx + q4] = v[e][q4];
            }
        } else {
            bf16x4 h;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) h[q4] = (__bf16)v[e][q4];
            *reinterpret_cast<bf16x4*>(S + idx) = h;                                                     // 8-B aligned in both images
        }
    }
}

// MFMA operand fragment of rows [row0, row0 + 32) for the k16 step `s` (bf16: 8 k values per lane = k 16 s + 8 h + 0..7)
template <bool ROWC, int ROWS> __device__ __forceinline__ bf16x8 frag_bf16(const __bf16* S, int row0, int s, int lane) {
    constexpr int P = Image<false, ROWC, ROWS>::pitch;
    if constexpr (!ROWC) {
        return *reinterpret_cast<const bf16x8*>(S + (row0 + (lane & 31)) * P + 16 * s + 8 * (lane >> 5));
    } else {
        // ds_read_b64_tr_b16: within a 16-lane group, lane c addresses the 8-byte chunk (k row c >> 2, feature block c & 3) of a
        // [4 k][16 rows] tile and receives column c of it, i.e. 4 consecutive k of its own row.  Group g: rows 16 (g & 1) + c,
        // k half g >> 1 — exactly the (l & 31, l >> 5) split of the MFMA operand.
        const int g = lane >> 4, c = lane & 15;
        const __bf16* q = S + (16 * s + 8 * (g >> 1) + (c >> 2)) * P + row0 + 16 * (g & 1) + 4 * (c & 3);
        union {
            s16x4 h[2];
            bf16x8 v;
        } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + 4 * P));
        return u.v;
    }
}
template <bool ROWC, int ROWS> __device__ __forceinline__ float frag_f32(const float* S, int row0, int k, int lane) {   // k = 2 s + h
    constexpr int P = Image<true, ROWC, ROWS>::pitch;
    if constexpr (!ROWC)
        return S[(row0 + (lane & 31)) * P + k];
    else
        return S[k * P + row0 + (lane & 31)];
}

// C[i][j] = sum_k A(i, k) B(j, k), a 128 x TJ tile per workgroup (TJ = 128: 4 waves as 2 x 2; TJ = 256: 8 waves as 2 x 4 — the A
// panel, i.e. the activations, is then read once for up to 256 output features).  ws != nullptr (weight gradient): the raw tile
// goes to ws[blockIdx.z][i][j] (whole tiles) and the row sums of A (the bias gradient) to gb_ws[blockIdx.z][i]; the reduce
// kernel finishes the job.  `tj` = number of j tiles; blockIdx.x = i tile * tj + j tile.
// VEC: both operands take 16-byte loads (the kernel without the scalar-load path).  Its bf16 instantiations are compiled for 4
// (TJ = 256: two 8-wave workgroups per CU, 13-17 spilled registers) / 3 (TJ = 128: three 4-wave workgroups) waves per SIMD: the
// kernel is bound by the bytes of register-staged loads in flight per CU, and the second workgroup buys 20 % (profiles/README.md).
template <bool F32, bool AROWC, bool BROWC, int TJ, bool VEC>
__global__ __launch_bounds__(TJ * 2, (!F32 && VEC) ? (TJ == 256 ? 4 : 3) : 1) void linear_gemm_kernel(Operand A, Operand B, int64_t K, int64_t k_per_split, float* __restrict__ C,
                                                             int64_t ldc, int64_t I, int64_t J, const float* __restrict__ bias, int act,
                                                             int accumulate, float* __restrict__ ws, float* __restrict__ gb_ws, int tj) {
    constexpr int TI = kTileI, NT = TJ * 2;
    using IA = Image<F32, AROWC, TI>;
    using IB = Image<F32, BROWC, TJ>;
    using UA = Units<AROWC, TI, NT>;
    using UB = Units<BROWC, TJ, NT>;
    using T = typename IA::T;
    __shared__ __attribute__((aligned(16))) T As[IA::elems];
    __shared__ __attribute__((aligned(16))) T Bs[IB::elems];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave & 1, wj = wave >> 1, h = lane >> 5, l31 = lane & 31;
    // linear workgroup id, j tile fastest: the workgroups that share an A panel are dispatched together
    const int64_t bi = blockIdx.x / (unsigned)tj, bj = blockIdx.x % (unsigned)tj;
    const int64_t i0 = bi * TI, j0 = bj * TJ;
    const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float rowsum[2] = {0.0f, 0.0f};
    const bool want_rowsum = gb_ws != nullptr && bj == 0 && wj == 0;

    // kDepth stages of global loads in flight per workgroup (register buffers): they travel under the current stage's MFMAs
    float va[kDepth][UA::UPT][4], vb[kDepth][UB::UPT][4];
    const int64_t nst = kbeg < kend ? (kend - kbeg + kStage - 1) / kStage : 0;
#pragma unroll
    for (int u = 0; u < kDepth; ++u)
        if (u < nst) {
            fetch<AROWC, TI, NT, VEC>(A, i0, kbeg + u * kStage, kend, va[u]);
            fetch<BROWC, TJ, NT, VEC>(B, j0, kbeg + u * kStage, kend, vb[u]);
        }
    for (int64_t st = 0; st < nst; st += kDepth) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            if (st + u >= nst) break;
            __syncthreads();                       // the previous stage's fragment reads are done
            stash<F32, AROWC, TI, NT>(As, va[u]);
            stash<F32, BROWC, TJ, NT>(Bs, vb[u]);
            __syncthreads();
            if (st + u + kDepth < nst) {
                fetch<AROWC, TI, NT, VEC>(A, i0, kbeg + (st + u + kDepth) * kStage, kend, va[u]);
                fetch<BROWC, TJ, NT, VEC>(B, j0, kbeg + (st + u + kDepth) * kStage, kend, vb[u]);
            }
            if constexpr (F32) {
#pragma unroll
                for (int s = 0; s < kStage / 2; ++s) {
                    const float a0 = frag_f32<AROWC, TI>(As, wi * 64, 2 * s + h, lane), a1 = frag_f32<AROWC, TI>(As, wi * 64 + 32, 2 * s + h, lane);
                    const float b0 = frag_f32<BROWC, TJ>(Bs, wj * 64, 2 * s + h, lane), b1 = frag_f32<BROWC, TJ>(Bs, wj * 64 + 32, 2 * s + h, lane);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                    if (want_rowsum) rowsum[0] = nh_add(rowsum[0], a0), rowsum[1] = nh_add(rowsum[1], a1);
                }
            } else {
#pragma unroll
                for (int s = 0; s < kStage / 16; ++s) {
                    const bf16x8 a0 = frag_bf16<AROWC, TI>(As, wi * 64, s, lane), a1 = frag_bf16<AROWC, TI>(As, wi * 64 + 32, s, lane);
                    const bf16x8 b0 = frag_bf16<BROWC, TJ>(Bs, wj * 64, s, lane), b1 = frag_bf16<BROWC, TJ>(Bs, wj * 64 + 32, s, lane);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
                    if (want_rowsum) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            rowsum[0] = nh_add(rowsum[0], (float)a0[e]), rowsum[1] = nh_add(rowsum[1], (float)a1[e]);
                    }
                }
            }
        }
    }
    // C/D fragment of v_mfma_f32_32x32x*: lane -> column (l & 31), register r -> row (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    if (ws) {
        const int64_t Ip = (int64_t)(gridDim.x / (unsigned)tj) * TI, Jp = (int64_t)tj * TJ;
        float* w = ws + ((int64_t)blockIdx.z * Ip) * Jp;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t i = i0 + wi * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    w[i * Jp + j0 + wj * 64 + b * 32 + l31] = acc[a][b][r];
                }
        if (want_rowsum) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float tot = nh_add(rowsum[a], __shfl_xor(rowsum[a], 32, 64));      // the two k halves of the operand
                if (h == 0) gb_ws[(int64_t)blockIdx.z * Ip + i0 + wi * 64 + a * 32 + l31] = tot;
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t j = j0 + wj * 64 + b * 32 + l31;
        if (j >= J) continue;
        const float bj_ = bias ? bias[j] : 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t i = i0 + wi * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (i >= I) continue;
                float* c = C + i * ldc + j;
                float v = acc[a][b][r];
                if (accumulate) v = nh_add(v, *c);
                if (bias) v = nh_add(v, bj_);
                if (act == NERFHIP_ACT_RELU) v = v < 0.0f ? 0.0f : v;                          // NaN stays NaN, as torch.relu
                else if (act == NERFHIP_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));            // as the fused kernel's rgb head
                *c = v;
            }
    }
}

// gw[i][j] (+)= sum_z ws[z][i][j];  gb[i] (+)= sum_z gb_ws[z][i]      (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void linear_dw_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ gb_ws, int splits,
                                                               int64_t Ip, int64_t Jp, int64_t I, int64_t Jw, float* __restrict__ gw,
                                                               int64_t ldgw, float* __restrict__ gb, int accumulate) {
    const int64_t Jx = Jw + (gb ? 1 : 0);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= I * Jx) return;
    const int64_t i = idx / Jx, j = idx - i * Jx;
    float s = 0.0f;
    if (j < Jw)
        for (int z = 0; z < splits; ++z) s = nh_add(s, ws[((int64_t)z * Ip + i) * Jp + j]);
    else
        for (int z = 0; z < splits; ++z) s = nh_add(s, gb_ws[(int64_t)z * Ip + i]);
    float* dst = (j < Jw) ? gw + i * ldgw + j : gb + i;
    *dst = accumulate ? nh_add(*dst, s) : s;
}

static int tile_j(int64_t J) {      // the 8-wave 128 x 256 tile when there is more than one 128-column tile
    static const char* force = getenv("NERFHIP_LIN_TJ");                        // A/B: NERFHIP_LIN_TJ=128
    if (force && force[0] == '1') return 128;
    return J > 128 ? 256 : 128;
}
static int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct DwPlan {
    int splits, TJ;
    int64_t k_per_split, Ip, Jp;
};
static DwPlan dw_plan(int64_t n, int n_in, int n_out) {
    DwPlan p;
    p.TJ = tile_j(n_in);
    const int64_t ti = cdiv(n_out, kTileI), tj = cdiv(n_in, p.TJ);
    p.Ip = ti * kTileI;
    p.Jp = tj * p.TJ;
    int64_t want = cdiv(1024, ti * tj * (p.TJ / 128));         // ~1024 four-wave workgroup equivalents (256 CUs x 4)
    const int64_t most = cdiv(n, 512);                          // at least 16 stages per workgroup
    if (want > most) want = most;
    if (want < 1) want = 1;
    int64_t per = cdiv(cdiv(n, want), kStage) * kStage;
    if (per < kStage) per = kStage;
    p.k_per_split = per;
    p.splits = (int)cdiv(n, per);
    if (p.splits < 1) p.splits = 1;
    return p;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// 16-byte loads: every pointer 16-byte aligned, every stride a whole number of float4s
static int vec_ok(const float* p, int64_t s, const float* y, int64_t ys) {
    return aligned16(p) && (s % 4 == 0) && (!y || (aligned16(y) && ys % 4 == 0));
}

// grid.x = (i tiles) x (j tiles), grid.z = splits
template <bool AROWC, bool BROWC, typename... Args>
static int launch_gemm(bool f32, int TJ, int64_t ti, int64_t tj, int splits, hipStream_t stream, const Operand& A, const Operand& B,
                       Args... args) {
    if (ti * tj > 0x7fffffff) return NERFHIP_E_BADARG;
    const dim3 grid((unsigned)(ti * tj), 1, (unsigned)splits);
    const bool vec = A.vec && B.vec;          // both operands 16-byte loadable: the kernel without the scalar-load path
#define NH_LIN_LAUNCH(F, T, V) hipLaunchKernelGGL((linear_gemm_kernel<F, AROWC, BROWC, T, V>), grid, dim3(2 * T), 0, stream, A, B, args..., (int)tj)
    if (TJ == 256) {
        if (f32) { if (vec) NH_LIN_LAUNCH(true, 256, true); else NH_LIN_LAUNCH(true, 256, false); }
        else     { if (vec) NH_LIN_LAUNCH(false, 256, true); else NH_LIN_LAUNCH(false, 256, false); }
    } else {
        if (f32) { if (vec) NH_LIN_LAUNCH(true, 128, true); else NH_LIN_LAUNCH(true, 128, false); }
        else     { if (vec) NH_LIN_LAUNCH(false, 128, true); else NH_LIN_LAUNCH(false, 128, false); }
    }
#undef NH_LIN_LAUNCH
    return nerfhip_launch_status();
}

}  // namespace lin
}  // namespace nerfhip

using nerfhip::lin::cdiv;
using nerfhip::lin::Operand;
using nerfhip::lin::vec_ok;

static bool lin_dtype_ok(int dtype) { return dtype == NERFHIP_F32 || dtype == NERFHIP_BF16 || dtype == NERFHIP_BF16_F8; }
static bool lin_act_ok(int act) { return act == NERFHIP_ACT_NONE || act == NERFHIP_ACT_RELU || act == NERFHIP_ACT_SIGMOID; }
static bool ld_ok(int64_t ld) { return ld > 0 && ld < (1 << 22); }   // a tile's element offsets r * ld + k (r < 256 rows) stay below 2^31 as 32-bit ints

extern "C" int nerfhip_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y,
                                  int64_t ldy, int64_t n, int n_in, int n_out, int act, int accumulate, int dtype,
                                  nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0 && n_in >= 1 && n_out >= 1 && ldx >= n_in && ldw >= n_in && ldy >= n_out);
    NERFHIP_CHECK_ARG(ld_ok(ldx) && ld_ok(ldw) && ld_ok(ldy));
    NERFHIP_CHECK_ARG(lin_dtype_ok(dtype) && lin_act_ok(act));
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(x && w && y);
    Operand A = {x, ldx, nullptr, 0, 0, n, vec_ok(x, ldx, nullptr, 0)};                   // A(i = point, k = in feature)
    Operand B = {w, ldw, nullptr, 0, 0, n_out, vec_ok(w, ldw, nullptr, 0)};               // B(j = out feature, k = in feature)
    const int TJ = nerfhip::lin::tile_j(n_out);
    return nerfhip::lin::launch_gemm<false, false>(dtype == NERFHIP_F32, TJ, cdiv(n, 128), cdiv(n_out, TJ), 1, (hipStream_t)stream, A, B,
                                                   (int64_t)n_in, (int64_t)n_in, y, ldy, n, (int64_t)n_out, bias, act, accumulate,
                                                   (float*)nullptr, (float*)nullptr);
}

extern "C" int nerfhip_linear_bwd_input(const float* gy, int64_t ldgy, const float* y, int64_t ldy, int act, const float* w,
                                        int64_t ldw, float* gx, int64_t ldgx, int64_t n, int n_in, int n_out, int accumulate,
                                        int dtype, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0 && n_in >= 1 && n_out >= 1 && ldgy >= n_out && ldw >= n_in && ldgx >= n_in);
    NERFHIP_CHECK_ARG(ld_ok(ldgy) && ld_ok(ldw) && ld_ok(ldgx) && ld_ok(ldy));
    NERFHIP_CHECK_ARG(lin_dtype_ok(dtype) && lin_act_ok(act) && (act == NERFHIP_ACT_NONE || (y && ldy >= n_out)));
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(gy && w && gx);
    const float* yy = act == NERFHIP_ACT_NONE ? nullptr : y;
    Operand A = {gy, ldgy, yy, ldy, act, n, vec_ok(gy, ldgy, yy, ldy)};                   // A(i = point, k = out feature), gated
    Operand B = {w, ldw, nullptr, 0, 0, n_in, vec_ok(w, ldw, nullptr, 0)};                // B(j = in feature, k = out feature) = w[k][j]
    const int TJ = nerfhip::lin::tile_j(n_in);
    return nerfhip::lin::launch_gemm<false, true>(dtype == NERFHIP_F32, TJ, cdiv(n, 128), cdiv(n_in, TJ), 1, (hipStream_t)stream, A, B,
                                                  (int64_t)n_out, (int64_t)n_out, gx, ldgx, n, (int64_t)n_in, (const float*)nullptr,
                                                  (int)NERFHIP_ACT_NONE, accumulate, (float*)nullptr, (float*)nullptr);
}

extern "C" size_t nerfhip_linear_bwd_weight_workspace_bytes(int64_t n, int n_in, int n_out) {
    if (n <= 0 || n_in < 1 || n_out < 1) return 0;
    const nerfhip::lin::DwPlan p = nerfhip::lin::dw_plan(n, n_in, n_out);
    return (size_t)p.splits * (size_t)p.Ip * (size_t)(p.Jp + 1) * sizeof(float);
}

extern "C" int nerfhip_linear_bwd_weight(const float* gy, int64_t ldgy, const float* y, int64_t ldy, int act, const float* x,
                                         int64_t ldx, float* gw, int64_t ldgw, float* gb, void* workspace, int64_t n, int n_in,
                                         int n_out, int accumulate, int dtype, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 1 && n_in >= 1 && n_out >= 1 && ldgy >= n_out && ldx >= n_in && ldgw >= n_in);
    NERFHIP_CHECK_ARG(ld_ok(ldgy) && ld_ok(ldx) && ld_ok(ldy));
    NERFHIP_CHECK_ARG(lin_dtype_ok(dtype) && lin_act_ok(act) && (act == NERFHIP_ACT_NONE || (y && ldy >= n_out)));
    NERFHIP_CHECK_ARG(gy && x && gw && workspace);
    const nerfhip::lin::DwPlan p = nerfhip::lin::dw_plan(n, n_in, n_out);
    const float* yy = act == NERFHIP_ACT_NONE ? nullptr : y;
    Operand A = {gy, ldgy, yy, ldy, act, n_out, vec_ok(gy, ldgy, yy, ldy)};               // A(i = out feature, k = point) = g[k][i]
    Operand B = {x, ldx, nullptr, 0, 0, n_in, vec_ok(x, ldx, nullptr, 0)};                // B(j = in feature, k = point) = x[k][j]
    float* ws = (float*)workspace;
    float* gb_ws = ws + (size_t)p.splits * p.Ip * p.Jp;
    int rc = nerfhip::lin::launch_gemm<true, true>(dtype == NERFHIP_F32, p.TJ, p.Ip / 128, p.Jp / p.TJ, p.splits, (hipStream_t)stream, A, B, n,
                                                   p.k_per_split, (float*)nullptr, (int64_t)0, (int64_t)n_out, (int64_t)n_in,
                                                   (const float*)nullptr, (int)NERFHIP_ACT_NONE, 0, ws, gb ? gb_ws : (float*)nullptr);
    if (rc) return rc;
    const int64_t total = (int64_t)n_out * (n_in + (gb ? 1 : 0));
    hipLaunchKernelGGL(nerfhip::lin::linear_dw_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, (const float*)gb_ws, p.splits, p.Ip, p.Jp, (int64_t)n_out, (int64_t)n_in, gw, ldgw, gb,
                       accumulate);
    return nerfhip_launch_status();
}
