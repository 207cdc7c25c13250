// K2b: backward of the fused NeRF MLP (autograd mirror of reference models/nerf.py:100-124 as driven by
// train.py:103-117 `loss.backward()`).  Two hand-written phases (A lives in mlp_bwd_chain.hip, B and the C ABI here):
//
//  A  mlp_bwd_chain  — per 32-point wave tile, the same register-resident chain as the forward, run in
//     reverse with W^T streamed through the LDS ring:  g_h(l-1) = W_l^T g_a(l),  g_a = g_h * relu'(h)
//     (masks read from the activations the forward saved).  Emits every dL/d(pre-activation) as slabs in
//     the forward's fragment order.  MFMA-bound, ~0.93x the forward's MFMA count.
//  B  mlp_bwd_dw     — dW_l = dY_l^T X_l with the POINTS as the MFMA K dimension.  dY/X tiles are DMA'd
//     global->LDS in fragment order (lane-linear, no address math) and transposed on the fly by
//     ds_read_b64_tr_b16 (bf16) so that lane = feature, regs = points; fp32 gathers with ds_read_b32.
//     A workgroup owns one (layer, point-range) job: wave w = output tile w against all X tiles, fp32
//     accumulators in registers for the whole range, one partial slab per workgroup; mlp_bwd_reduce sums
//     the slabs, un-permutes features and writes the (out,in) gradient tensors + biases.
//     HBM-bound by construction: 2*256*256 FLOP per 2*256*2 B = 128 FLOP/B.
//  C  mlp_bwd_fold   — xyz_encoding_final is a linear layer without activation: its saved input / output gradient are not needed
//     (mlp_layout.h kDwJobs).  The dir job forms G = dY_dir^T h8 instead of dY_dir^T f; this small fp32 kernel finishes
//     dW_dir[:, :256] = G W_f^T + s b_f^T,  dW_final = W_dx^T G,  db_final = W_dx^T s  from G, s = db_dir and the fp32 fold block
//     of the packed W^T image.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "mlp_layout.h"
#include "f8_store.h"
#include "adam_math.h"
#include "mlp_bwd_chain.h"

#ifndef NERFHIP_STORE_AUX
#define NERFHIP_STORE_AUX 2  // cache-policy bits of the dY stores: 2 = nt (-7 %; whole training step 1.65 -> 1.51 ms)
#endif
#ifndef NERFHIP_DW_NT
#define NERFHIP_DW_NT 1      // non-temporal LDS-DMA loads in the dW kernel (every byte is read once): 508 -> 466 us
#endif

namespace nerfhip {
using namespace mlp;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void glds16b(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
// same, non-temporal: for streams every byte of which is read once (the dW kernel's dY / X slabs)
__device__ __forceinline__ void glds16b_nt(const void* gsrc, unsigned lds_dst) {
#if NERFHIP_DW_NT
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
#else
    glds16b(gsrc, lds_dst);
#endif
}


// the same for the lanes of `mask` only (wave-uniform): the LDS image is lane-linear, so the other lanes' 16-byte units are simply not
// fetched; the instruction still counts once in vmcnt
__device__ __forceinline__ void glds16b_nt_masked(const void* gsrc, unsigned lds_dst, unsigned long long mask) {
    unsigned keep;
    unsigned long long keep_exec;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, %4\n\tglobal_load_lds_dwordx4 %2, off nt\n\t"
        "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(keep_exec)
        : "v"(gsrc), "s"(lds_dst), "s"(mask)
        : "memory");
}

// s_waitcnt vmcnt(N) + s_barrier, N a compile-time constant of the (job class) loop it sits in
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
// wall clock in 10 ns ticks (s_memrealtime, 100 MHz) — the probe builds' time base
__device__ __forceinline__ unsigned shader_cycles() { return (unsigned)__builtin_amdgcn_s_memrealtime(); }
#ifndef NERFHIP_DW_PROBE
#define NERFHIP_DW_PROBE 0       // debug builds: every wave of the dW kernels accumulates where its cycles go (tools/dw_probe.py)
#endif
#if NERFHIP_DW_PROBE
__device__ unsigned g_dw_probe[1024 * 8 * 8];       // [workgroup][wave][iters, wait, barrier, issue, compute, total, job, depth]
#endif

// ================================================================================================
// Phase B: weight gradients
// ================================================================================================
// Jobs + their point-range splits.  A workgroup = (job, split); how many splits a job gets is the host's plan (dw_plan below).
// One launch serves up to kDwMaxModels models (a training step's fine and coarse network): job j belongs to model j / 12 and
// carries that model's tensors, so ONE dW launch and ONE reduce launch cover the whole step.
constexpr int kDwMaxModels = 2;
constexpr int kDwMaxJobs = kDwMaxModels * kNumDwJobs;
struct DwJobTable {
    DwJob job[kDwMaxJobs];
    int nsplit[kDwMaxJobs];
    int soff[kDwMaxJobs + 1];     // prefix sums: workgroup / partial-slab index of (job j, split 0); = total for j >= njobs
    const uint8_t* acts[kDwMaxJobs];   // saved activations of the job's model
    const uint8_t* dys[kDwMaxJobs];    // dY slabs of the job's model
    int64_t ntiles[kDwMaxJobs];        // 32-point wave tiles of the job's model
    int njobs;
    // The sigma head's job has no workgroups of its own — its X (h8: 16 slabs per tile) is also the second X section of the DIR job
    // since round 6 (mlp_layout.h kDwJobs; rounds 4-5: of the final layer's job) — so the dir job's workgroups also form dW_sigma:
    // their waves 4..7, idle otherwise (the job has 4 dY tiles), each multiply dY_sigma by two of the eight h8 tiles.
    // fold_of[sigma job] = the dir job's index (its partial slabs hold the sigma partials in rows 4..7, which it does not use),
    // -1 everywhere else.
    int fold_of[kDwMaxJobs];
    // Encodings regenerated instead of read (bf16, nerfhip_mlp_bwd_multi_rays): per MODEL the rays (B,8), the depths (B,S) its forward
    // ran on and S / 32 (tiles per ray); enc_rays[m] == nullptr: the job's x1 section is read from the saved activations as before.
    const float* enc_rays[kDwMaxModels];
    const float* enc_z[kDwMaxModels];
    int enc_tpr[kDwMaxModels];
};
static_assert(kDwJobs[kDwJobDir].x2_off == kDwJobs[kDwJobSigma].x1_off && kDwJobs[kDwJobDir].x2_slabs == 16 && kDwJobs[kDwJobDir].x1_slabs == 2 &&
              kDwJobs[kDwJobSigma].x1_slabs == 16 && kDwJobs[kDwJobSigma].dy_slabs == 2 && kDwJobs[kDwJobDir].dy_slabs == 8 &&
              kDwJobs[kDwJobSigma].x2_slabs == 0, "the sigma head and the dir layer read the same h8 section");
// the folded sigma head in a dir-job partial slab: wave w = 4..7 holds (dY_sigma tile 0) x (h8 tiles 2 (w - 4), 2 (w - 4) + 1) in
// blocks (row w, columns 0, 1); its bias partial is bias row kDwFoldRow0 (written by wave 4)
constexpr int kDwFoldRow0 = 4;
NH_HD constexpr int dw_fold_block(int xt) { return (kDwFoldRow0 + xt / 2) * kDwMaxXTiles + (xt & 1); }     // block index of h8 tile xt
constexpr int kDwFoldStageSlabs = 28;  // [dY_dir 8][enc_dir 2][h8 16][dY_sigma 2]
constexpr int kDwFoldSigmaSlab = 26;   // first dY_sigma slab of that stage
#ifndef NERFHIP_DW_RING_KB
#define NERFHIP_DW_RING_KB 160   // bf16 dW ring: the whole LDS of a CU, cut into as many stages as the JOB's stage size allows (round 4; the
                                 // depth itself measured neutral — 4 stages of 36 KiB run the same 480 us — the waves never wait for data)
#endif
#ifndef NERFHIP_DW_MAXDEPTH
#define NERFHIP_DW_MAXDEPTH 12
#endif
#ifndef NERFHIP_DW_SHARE_LAST
#define NERFHIP_DW_SHARE_LAST 1  // bf16: the waves share a stage's last REM < 8 pieces under EXEC masks (0 = the surplus waves re-fetch the last piece)
#endif
#ifndef NERFHIP_DW_RD
#define NERFHIP_DW_RD 5          // bf16: B fragments in flight (ring of RD, RD - 1 steps ahead of the MFMA)
#endif
#ifndef NERFHIP_DW_SPREAD
#define NERFHIP_DW_SPREAD 1      // bf16: the next stage's DMAs issued between the current stage's MFMAs (0 = in one block after the barrier)
#endif

#ifndef NERFHIP_DW_SPLIT2D
#define NERFHIP_DW_SPLIT2D 1     // bf16, jobs with 8 dY tiles and 8 / 10 X tiles: wave = 2 dY tiles x (4 | 5) X tiles instead of 1 x (8 | 10) — 6 | 7
#endif                           // operand fragments from LDS per k-step instead of 9 | 11 (round 6: profiles/r06_dw_bisect.txt, variant M)
#ifndef NERFHIP_DW_RD2
#define NERFHIP_DW_RD2 4         // ... its B fragments in flight
#endif
#ifndef NERFHIP_DW_BIAS_DOT2
#define NERFHIP_DW_BIAS_DOT2 1   // bf16 bias partials by v_dot2_f32_bf16 against (1, 1), two chains, instead of 8 dependent cvt + add per fragment
#endif

#ifndef NERFHIP_DW_WGS
#define NERFHIP_DW_WGS 512       // target workgroup count of the fp32 dW launch (2 rounds of 256 CUs at 1 workgroup/CU)
#endif
#ifndef NERFHIP_DWBF16_WGS
#define NERFHIP_DWBF16_WGS 256   // bf16: ONE round.  With the pipelined inner loop the kernel itself is as fast in one round as in two
#endif                           // (585 vs 591 us merged), and every workgroup less is a 330 KB partial slab not written and not
                                 // re-read by the reduce: the bf16 step 1.265 -> 1.196 ms on the same box

template <int PREC> struct DwTraits;
template <> struct DwTraits<NERFHIP_BF16> {
    static constexpr int SPP = 1;            // 1 KiB pieces per slab
    static constexpr int RING_BYTES = NERFHIP_DW_RING_KB * 1024;      // cut into stages of the job class's own size (dw_depth)
    static constexpr int DEPTH = 0, STAGE_BYTES = 0;                  // (fp32 only: a fixed 2 x 72 KiB ring)
};
template <> struct DwTraits<NERFHIP_F32> {
    static constexpr int SPP = 2;
    static constexpr int DEPTH = 2;
    static constexpr int MAXP = 72;
    static constexpr int STAGE_BYTES = MAXP * kPieceBytes;
    static constexpr int RING_BYTES = DEPTH * STAGE_BYTES;
};

// ring stages of a job class whose stage is `pieces` KiB
template <int PREC> NH_HD constexpr int dw_depth(int pieces) {
    if (PREC != NERFHIP_BF16) return DwTraits<PREC>::DEPTH;
    const int d = DwTraits<PREC>::RING_BYTES / (pieces * kPieceBytes);
    return d > NERFHIP_DW_MAXDEPTH ? NERFHIP_DW_MAXDEPTH : d;
}

// One (dY tile, X tile) block of a workgroup's partial slab: 1024 floats, REGISTER-major since round 6 — float4 q of lane l at
// float4 index 64 q + l, so that every store instruction of the epilogue writes 1 KiB contiguous (lane-major, a lane's 16 floats
// together, made each of them 64 separate 16-byte pieces: profiles/r06_dw_bisect.txt, variants 7 -> F).  mlp_bwd_reduce_kernel
// decodes the same order.
__device__ __forceinline__ void dw_store_block(float* __restrict__ block, const f32x16& a, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        reinterpret_cast<float4*>(block)[64 * q + lane] = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}

// sum of a bf16 A fragment's 8 values into two running fp32 partials (the bias gradient: dY summed over the points)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ void dw_bias_sum(const bf16x8& a, float& s0, float& s1) {
#if NERFHIP_DW_BIAS_DOT2
    const bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f};
    s0 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{a[0], a[1]}, one, s0, false);
    s1 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{a[2], a[3]}, one, s1, false);
    s0 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{a[4], a[5]}, one, s0, false);
    s1 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{a[6], a[7]}, one, s1, false);
#else
#pragma unroll
    for (int j = 0; j < 8; ++j) s0 += (float)a[j];
#endif
}

// 4 bytes per lane, global -> LDS (lane L lands at lds_dst + 4 L)
__device__ __forceinline__ void glds4b_dw(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// Slab `ks` (wave-uniform) of the F-frequency encoding of v in the forward's slot order — the arithmetic of the bf16 forward's
// encode_slots (mlp_fwd_kernel.h, NERFHIP_FAST_SINCOS path: x / 2 pi as a hi + lo pair, exact power-of-two scaling, v_fract, hardware
// v_sin / v_cos in revolutions), operation for operation, so that the regenerated operand has the bits the forward multiplied by:
// pair p = 4 ks + q is channel p % 3 at frequency 2^(2 (p / 3) + h); the slots behind the last pair hold the identity channels.
template <int F, int SLABS>
__device__ __forceinline__ bf16x8 dw_encode_slab(const float (&v)[3], int h, int ks) {
    constexpr int NPAIR = 3 * (F / 2);
    constexpr float kInv2PiHi = 0.15915494f, kInv2PiLo = 6.4206297e-9f;
    float rh[3], rl[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float vs = h ? 2.0f * v[c] : v[c];
        rh[c] = vs * kInv2PiHi;
        rl[c] = __builtin_fmaf(vs, kInv2PiHi, -rh[c]) + vs * kInv2PiLo;
    }
    bf16x8 out;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int p = 4 * ks + q;
        float s, co;
        if (p < NPAIR) {
            const int i = p / 3, c = p - 3 * i;
            const float sc = (float)(1 << (2 * i));
            const float rhc = c == 0 ? rh[0] : (c == 1 ? rh[1] : rh[2]), rlc = c == 0 ? rl[0] : (c == 1 ? rl[1] : rl[2]);
            const float t = __builtin_amdgcn_fractf(rhc * sc) + rlc * sc;
            s = __builtin_amdgcn_sinf(t);
            co = __builtin_amdgcn_cosf(t);
        } else {
            const int tail = 2 * (p - NPAIR);
            s = (tail == 0) ? (h ? v[2] : v[0]) : 0.0f;
            co = (tail == 0) ? (h ? 0.0f : v[1]) : 0.0f;
        }
        out[2 * q] = (__bf16)s;
        out[2 * q + 1] = (__bf16)co;
    }
    return out;
}

template <int PREC>
__global__ __launch_bounds__(512, 2)
void mlp_bwd_dw_kernel(DwJobTable jobs, float* __restrict__ slabs) {
    constexpr int SPP = DwTraits<PREC>::SPP;
    constexpr int SLAB_BYTES = SPP * kPieceBytes;
    constexpr int IL = act_il(PREC);
    // The ring is sized in BYTES, not stages (round 4): a stage of a job is its own (dY + X slabs) KiB, and the ring holds as many
    // of them as fit — bf16: 4 for the skip layer (36 KiB), 5 for the 256 x 256 layers, 6 / 8 / 8 / 12 for the dir / first / sigma /
    // rgb jobs.  Job class = (X tiles, slabs per stage): the iteration loop exists once per class, so the stage count, the DMAs
    // per wave and the counted vmcnt of its wait are compile-time constants.
    constexpr int RING_BYTES = DwTraits<PREC>::RING_BYTES;
    __shared__ __attribute__((aligned(1024))) char ring[RING_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int jid = 0;
#pragma unroll
    for (int j = 1; j < kDwMaxJobs; ++j) jid += ((int)blockIdx.x >= jobs.soff[j]) ? 1 : 0;
    const int nsplit = jobs.nsplit[jid], split = (int)blockIdx.x - jobs.soff[jid];
    const DwJob jb = jobs.job[jid];
    const int64_t ntiles = jobs.ntiles[jid];
    const uint8_t* __restrict__ acts_base = jobs.acts[jid];
    const uint8_t* __restrict__ dys_base = jobs.dys[jid];
    [[maybe_unused]] const float* __restrict__ enc_rays = jobs.enc_rays[jid / kNumDwJobs];      // non-null: regenerate the encodings
    [[maybe_unused]] const float* __restrict__ enc_z = jobs.enc_z[jid / kNumDwJobs];
    [[maybe_unused]] const int enc_tpr = jobs.enc_tpr[jid / kNumDwJobs];
    const int n_ot = jb.dy_slabs / 2;
    const int n_xs = jb.x1_slabs + jb.x2_slabs;
    const int n_xt = n_xs / 2;
#ifndef NERFHIP_DW_BLOCKED
#define NERFHIP_DW_BLOCKED 1
#endif
#if NERFHIP_DW_BLOCKED
    // contiguous tile range per split (consecutive iterations stay inside the same 2 MiB pages: a tile block is
    // 167 KiB; the strided assignment touched 2-3 new pages per iteration per workgroup)
    const int64_t per = (ntiles + nsplit - 1) / nsplit;
    const int64_t t_first = (int64_t)split * per;
    const int64_t my_tiles = (t_first >= ntiles) ? 0 : ((ntiles - t_first < per) ? ntiles - t_first : per);
#else
    const int64_t t_first = split;
    const int64_t my_tiles = (ntiles - split + nsplit - 1) / nsplit;   // tiles split, split+nsplit, ...
#endif
    const unsigned lds_base = (unsigned)(uintptr_t)ring;

    // stage image: [dy slabs][x1 slabs][x2 slabs], each slab SPP 1 KiB pieces at 1 KiB pitch.  The DMA writes
    // LDS lane-linearly (16-B unit L of a piece <- lane L) but each lane chooses WHICH global 16-B unit it
    // fetches: bf16 pieces are stored in HBM as [half h][point n] and land in LDS as unit (2n+h) for even slabs
    // and (2n+h)^8 for odd slabs, so that the 32 lanes of a ds_read_b64_tr_b16 group (4 points x 2 halves x
    // 2 slabs x 2 j-halves) hit 32 distinct bank pairs.  (The linear [h][n] image was 4-way conflicted: the h,
    // slab and k-step strides are all multiples of 256 B.)
    const int dma_off_even = (PREC == NERFHIP_BF16) ? ((lane & 1) * 32 + (lane >> 1)) * 16 : lane * 16;
    const int dma_off_odd = (PREC == NERFHIP_BF16) ? ((lane & 1) * 32 + ((lane ^ 8) >> 1)) * 16 : lane * 16;

    f32x16 acc[kDwMaxXTiles];
#pragma unroll
    for (int x = 0; x < kDwMaxXTiles; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;
    float dbacc = 0.0f;
    [[maybe_unused]] float dbacc2 = 0.0f;             // (second chain of the dot2 bias sums)

    // per-lane transposing-read geometry (bf16): 16-lane group g reads a [4 points][16 features] tile whose
    // 8-byte chunks are (point row = c>>2, feature block = c&3) of lane c; feature block b lives in half
    // h=b&1, j-half b>>1 of the slab image  [h][point n][8 x bf16].
    const int grp = lane >> 4, c = lane & 15;
    // image unit of (point n, half h) = (2n + h) ^ (8 * slab parity);  n = 8*(grp>>1) + (c>>2) + 4*s + 16*q
    // (s = second read of the k-step, q = k-step): bit 3 of the unit is s, so odd slabs swap the two reads.
    const int tr_off = (grp & 1) * SLAB_BYTES + (2 * (8 * (grp >> 1) + (c >> 2)) + ((c & 3) & 1)) * 16 + ((c & 3) >> 1) * 8;
    const int tr_s0 = (grp & 1) ? 128 : 0, tr_s1 = 128 - tr_s0;
    // fp32 gather geometry: lane (m = l&31, k = l>>5): feature m -> slab m>>4, natural i = m&15 -> (h,j)
    const int m32 = lane & 31, kk = lane >> 5;
    const int f32_off = (m32 >> 4) * SLAB_BYTES + (slab_nat_h(m32 & 15) * 32) * 32 + slab_nat_j(m32 & 15) * 4;
#if NERFHIP_DW_PROBE
    unsigned pr_wait = 0, pr_bar = 0, pr_issue = 0, pr_comp = 0, pr_depth = 0;
    const uint64_t pr_t00 = __builtin_amdgcn_s_memrealtime();
#endif

    // One copy of the iteration loop per job class (X tiles NXT, slabs per stage NSL): straight-line X loop with the next tiles'
    // LDS reads in flight under the current tile's MFMA (see mlp_bwd_dw_f8_kernel), LPW = ceil(pieces / 8) DMAs per wave per stage
    // (the surplus of the last round re-fetches the stage's last piece: every wave issues the SAME count, so one immediate
    // vmcnt serves all), D ring stages.
    auto run = [&](auto nxt_c, auto nsl_c, auto regen_c) {
        constexpr int NXT = decltype(nxt_c)::value, NSL = decltype(nsl_c)::value;
        // REGEN (bf16; classes whose x1 section is an input encoding: first layer, skip layer, dir layer): the ENC encoding slabs of a
        // stage are not fetched — waves 0 .. ENC - 1 form one slab each from the tile's depths (128 B, DMA'd one stage AHEAD of the
        // stage's pieces into a small ring behind the stages) and the ray (scalar loads), and write it where the DMA would have put it
        constexpr bool REGEN = decltype(regen_c)::value;
        static_assert(!REGEN || (PREC == NERFHIP_BF16 && (NXT == 2 || NXT == 9 || NXT == 10)), "classes with an encoding section");
        constexpr int ENC = REGEN ? (NXT == 9 ? kDirSlabs : kXyzSlabs) : 0;
        constexpr int DYS = NXT == 9 ? 8 : 16;                     // (REGEN) dY slabs ahead of the encoding section in the stage image
        // class (9, 28) = the dir layer with the sigma head folded in: stage = [dY_dir 8][enc_dir 2][h8 16][dY_sigma 2] slabs; the
        // waves 4..7 (no dY tile of the dir layer is theirs) multiply dY_sigma by the h8 tiles 2 (w - 4), 2 (w - 4) + 1
        constexpr bool FOLD = NXT == 9 && NSL == kDwFoldStageSlabs;
        // round 6, bf16: the classes with 8 dY tiles and 8 | 10 X tiles — (8, 32), (10, 36): 85 % of the launch's
        // workgroups — give wave (wi = wave >> 1, wj = wave & 1) the dY tiles 2 wi, 2 wi + 1 against the X tiles XW wj .. XW wj + XW - 1
        constexpr bool SPLIT2D = (PREC == NERFHIP_BF16) && NERFHIP_DW_SPLIT2D && (NXT == 8 || NXT == 10) && (NSL - 2 * NXT >= 16);
        constexpr int NP = NSL * SPP;                              // 1 KiB pieces per stage
        constexpr int NPD = NP - ENC;                              // ... of which are fetched
        constexpr int LPWD = (NPD + 7) / 8;                        // piece DMAs per wave per stage
        constexpr int LPW = LPWD + (REGEN ? 1 : 0);                // + the next stage's depths (every wave: one vmcnt immediate for all)
        constexpr int STAGE = (PREC == NERFHIP_BF16) ? NP * kPieceBytes : DwTraits<PREC>::STAGE_BYTES;
        constexpr int ZSLOT = 256;                                 // bytes of one stage's depths in LDS: 64 lanes x 4 B (lanes 32.. repeat)
        constexpr int D0 = dw_depth<PREC>(NP);
        constexpr int D = (REGEN && D0 * (STAGE + ZSLOT) > RING_BYTES) ? D0 - 1 : D0;
        static_assert(D >= 2 && D * (STAGE + (REGEN ? ZSLOT : 0)) <= RING_BYTES, "ring stages of this job class");
        static_assert((D - 2) * LPW <= 63, "counted vmcnt");
#if NERFHIP_DW_PROBE
        pr_depth = D;
#endif
        int s_issue = 0, s_use = 0;               // ring slots of the next stage to fetch / to consume (wave-uniform, wrap at D)
        // the stage a fetch goes to: tile block pointers + ring slot (wave-uniform), then one DMA per piece
        const uint8_t* abase = nullptr;
        const uint8_t* dbase = nullptr;
        unsigned slot = 0;
        auto stage_tile = [&](int64_t it) {
            int64_t T = t_first + (it < my_tiles ? it : my_tiles - 1) * (NERFHIP_DW_BLOCKED ? 1 : nsplit);   // past the end: re-fetch
            if (T >= ntiles) T = ntiles - 1;
            return T;
        };
        // (REGEN) depths of stage `st`: ring of D slots behind the stages, slot = st mod D
        const float* zsrc = nullptr;
        unsigned zslot = 0;
        int z_issue = 0, g_slot = 0;
        const unsigned lds_z = lds_base + (unsigned)(D * STAGE);
        auto next_z = [&](int64_t st) {
            zsrc = enc_z + stage_tile(st) * 32 + (lane & 31);
            zslot = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_z + (unsigned)(z_issue * ZSLOT)));
            z_issue = (z_issue + 1 == D) ? 0 : z_issue + 1;
        };
        auto next_stage = [&](int64_t it) {
            const int64_t T = stage_tile(it);
            abase = acts_base + tile_block_off(T, act_tile_bytes(PREC), IL);      // (bf16: the block's pieces are IL KiB apart, mlp_layout.h)
            dbase = dys_base + tile_block_off(T, kDySlabs * 64 * (16 * SPP), IL);
            slot = lds_base + (unsigned)(s_issue * STAGE);
            s_issue = (s_issue + 1 == D) ? 0 : s_issue + 1;
            if constexpr (REGEN) next_z(it + 1);
        };
        // (REGEN) the encoding slab `wave` of stage `st` (ring slot g_slot, depths in z slot g_slot) written into the stage image in the
        // unit order the DMA gives the fetched slabs: (point n, half h) -> unit (2 n + h) ^ (8 x slab parity)
        // The ray (origin, direction) of a stage's tile by SCALAR loads (constant address space, wave-uniform address), fetched one
        // iteration before it is used: a vector load in the loop makes hipcc drain vmcnt — the whole DMA ring — every iteration
        // (measured: the launch 400 -> 600 us), and a scalar load issued where it is needed puts a memory round trip into the
        // iteration of every generating wave, hence — one barrier per stage — of the workgroup (430 us).
        float ray_next[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto load_ray = [&](int64_t st) {
            if (wave < ENC) {
                const unsigned r = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)stage_tile(st) / (unsigned)enc_tpr));
                typedef const float __attribute__((address_space(4))) * ConstF;
                const ConstF rp = (ConstF)(uintptr_t)(enc_rays + (size_t)r * 8);
#pragma unroll
                for (int c = 0; c < 6; ++c) ray_next[c] = rp[c];
            }
        };
        auto gen_stage = [&](int64_t st) {
            if (wave < ENC) {
                const int n = lane & 31, h = lane >> 5;
                bf16x8 e;
                if constexpr (NXT == 9) {
                    const float dv[3] = {ray_next[3], ray_next[4], ray_next[5]};
                    e = dw_encode_slab<4, kDirSlabs>(dv, h, wave);
                } else {
                    const float zv = *reinterpret_cast<const float*>(ring + D * STAGE + g_slot * ZSLOT + n * 4);
                    float xv[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) xv[c] = nh_add(ray_next[c], nh_mul(ray_next[3 + c], zv));      // o + d z   rendering.py:206-207
                    e = dw_encode_slab<10, kXyzSlabs>(xv, h, wave);
                }
                const int unit = (2 * n + h) ^ ((wave & 1) ? 8 : 0);
                *reinterpret_cast<bf16x8*>(ring + g_slot * STAGE + (DYS + wave) * SLAB_BYTES + unit * 16) = e;
            }
            g_slot = (g_slot + 1 == D) ? 0 : g_slot + 1;
            load_ray(st + 1);
        };
        // The last of a wave's LPW DMAs per stage: when the stage has REM = NP mod 8 pieces left for it (4 or 2 in the bf16 classes), the
        // 8 waves SHARE them — 8 / REM waves per piece, each fetching its 64 REM / 8 lanes' units under an EXEC mask — instead of
        // 8 - REM waves re-fetching the stage's last piece: every wave still issues the same count (one immediate vmcnt), and no byte
        // is fetched twice (round 4: the re-fetches were up to a quarter of a small job's DMAs, and nt loads do not stay in L2).
        constexpr int REM = NPD % 8;
        constexpr bool SHARE_LAST = (PREC == NERFHIP_BF16) && NERFHIP_DW_SHARE_LAST && (REM == 4 || REM == 2);
        const int share_piece = NPD - REM + (SHARE_LAST ? (wave * REM) / 8 : 0);
        const unsigned long long share_mask = REM == 4 ? (0xffffffffull << (32 * (wave & 1))) : (0xffffull << (16 * (wave & 3)));
        auto issue_piece = [&](int i) {
            if constexpr (REGEN) {
                if (i == LPW - 1) {                                             // the NEXT stage's depths
                    glds4b_dw(zsrc, zslot);
                    return;
                }
            }
            int pi = wave + 8 * i;
            const bool shared = SHARE_LAST && i == LPWD - 1;
            if (shared) pi = share_piece;
            if (pi >= NPD) pi = NPD - 1;                                        // (classes without sharing) duplicate DMA of the last piece
            if (REGEN && pi >= DYS) pi += ENC;                                  // fetched piece -> piece of the stage image
            const int sl = pi / SPP, sub = pi % SPP;
            const uint8_t* src;
            if (FOLD && sl >= kDwFoldSigmaSlab) src = dbase + (size_t)(kDySigma + sl - kDwFoldSigmaSlab) * 64 * (16 * SPP) * IL;
            else if (sl < jb.dy_slabs) src = dbase + (size_t)(jb.dy_off + sl) * 64 * (16 * SPP) * IL;
            else if (sl < jb.dy_slabs + jb.x1_slabs) src = abase + (size_t)(jb.x1_off + sl - jb.dy_slabs) * 64 * (16 * SPP) * IL;
            else src = abase + (size_t)(jb.x2_off + sl - jb.dy_slabs - jb.x1_slabs) * 64 * (16 * SPP) * IL;
            // fp32: a slab is 64 lanes x 32 B; piece `sub` = lanes' bytes [16*sub, 16*sub+16) is NOT contiguous,
            // so DMA whole 1 KiB lines instead: line q of the slab = lanes 32q..32q+31 (32 B each).
            const uint8_t* g = src + (size_t)sub * kPieceBytes + ((sl & 1) ? dma_off_odd : dma_off_even);
            // (the destination is wave-uniform; said so, because hipcc otherwise shares a VGPR copy of pi x 1 KiB with the source address)
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(slot + (unsigned)(pi * kPieceBytes)));
            if (shared) glds16b_nt_masked(g, dst, share_mask);
            else glds16b_nt(g, dst);
        };
        auto issue_stage = [&](int64_t it) {
            next_stage(it);
#pragma unroll
            for (int i = 0; i < LPW; ++i) issue_piece(i);
        };
        // bf16: the next stage's DMAs are issued one by one BETWEEN the MFMAs of the current stage (round 4).  Issued in one block at
        // the top of the iteration — all 8 waves at once — they are 32-36 KiB through the CU's 64 B/clk texture-address path:
        // tools/dw_probe.py measured 0.35 us of every 1.5 us iteration in that block, 0.27 us at the barrier behind it and no time
        // at all waiting for data.  One DMA every DMA_STEP MFMAs hides the path's back-pressure under the other wave's MFMAs.
        constexpr bool SPREAD = (PREC == NERFHIP_BF16) && NERFHIP_DW_SPREAD;
        constexpr int DMA_STEP = (2 * NXT) / LPW > 0 ? (2 * NXT) / LPW : 1;
        if constexpr (REGEN) {                    // stage 0's depths lead the queue
            next_z(0);
            glds4b_dw(zsrc, zslot);
            load_ray(0);
        }
#pragma unroll
        for (int s = 0; s < D - 1; ++s) issue_stage(s);
        if constexpr (REGEN) {
            wait_vm<(D - 1) * LPW>();             // (this wave's copy of) stage 0's depths landed; every wave fetched the same 128 B
            gen_stage(0);
        }
        for (int64_t it = 0; it < my_tiles; ++it) {
            // stage `it` landed (D-2 younger stages may still fly), everyone done with stage it-1
#if NERFHIP_DW_PROBE
            const unsigned t0 = shader_cycles();
            wait_vm<(PREC == NERFHIP_BF16) ? (D - 2) * LPW : 0>();
            const unsigned t1 = shader_cycles();
            asm volatile("s_barrier" ::: "memory");
            const unsigned t2 = shader_cycles();
            pr_wait += (t1 - t0) & 0xffffffffu;
            pr_bar += (t2 - t1) & 0xffffffffu;
#else
            wait_vm_barrier<(PREC == NERFHIP_BF16) ? (D - 2) * LPW : 0>();
#endif
            if (SPREAD && (SPLIT2D || wave < n_ot)) next_stage(it + D - 1);
            else issue_stage(it + D - 1);
#if NERFHIP_DW_PROBE
            const unsigned t3 = shader_cycles();
            pr_issue += (t3 - t2) & 0xffffffffu;
#endif
            const char* st_base = ring + s_use * STAGE;
            s_use = (s_use + 1 == D) ? 0 : s_use + 1;
            if constexpr (SPLIT2D) {
                constexpr int XW = NXT / 2, NF = 2 * XW, RD = NERFHIP_DW_RD2;
                static_assert(RD >= 2 && RD <= NF, "B fragment ring");
                const int wi = wave >> 1, wj = wave & 1;
                const char* dyb = st_base + (4 * wi) * SLAB_BYTES;                       // dY tiles 2 wi, 2 wi + 1
                const char* x_all = st_base + 16 * SLAB_BYTES;
                const char* xb = x_all + (2 * XW * wj) * SLAB_BYTES;                    // X tiles XW wj ..
                auto load_frag = [&](const char* pb, int q) {
                    union { s16x4 h2[2]; bf16x8 v; } f;
                    f.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + tr_off + q * 512 + tr_s0));
                    f.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + tr_off + q * 512 + tr_s1));
                    return f.v;
                };
                // piece k of the next stage is issued behind B step dma_after(k): the LPW DMAs spread evenly over the NF steps
                auto dma_after = [](int k) constexpr { return ((k + 1) * NF) / LPW - 1; };
                bf16x8 a[2][2], b[RD];
                a[0][0] = load_frag(dyb, 0);
                a[0][1] = load_frag(dyb + 2 * SLAB_BYTES, 0);
#pragma unroll
                for (int f = 0; f < RD - 1; ++f) b[f] = load_frag(xb + 2 * (f % XW) * SLAB_BYTES, f / XW);
                a[1][0] = load_frag(dyb, 1);
                a[1][1] = load_frag(dyb + 2 * SLAB_BYTES, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF; ++f) {        // B step f = (k-step f / XW, X tile f % XW of this wave): two MFMAs
                    if (f + RD - 1 < NF) b[(f + RD - 1) % RD] = load_frag(xb + 2 * ((f + RD - 1) % XW) * SLAB_BYTES, (f + RD - 1) / XW);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[2 * (f % XW)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[f / XW][0], b[f % RD], acc[2 * (f % XW)], 0, 0, 0);
                    acc[2 * (f % XW) + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[f / XW][1], b[f % RD], acc[2 * (f % XW) + 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (f == 0 || f == XW) {          // bias partial of dY tile 2 wi + wj (its two waves share the pair's two tiles)
                        const bf16x8 ab = wj ? a[f / XW][1] : a[f / XW][0];
                        dw_bias_sum(ab, dbacc, dbacc2);
                    }
                    if constexpr (SPREAD) {
#pragma unroll
                        for (int k = 0; k < LPW; ++k)
                            if (dma_after(k) == f) {
                                issue_piece(k);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
                }
            } else if (wave < n_ot) {
                const char* dy_base = st_base + (2 * wave) * SLAB_BYTES;
                const char* x_base = st_base + jb.dy_slabs * SLAB_BYTES;
                if constexpr (PREC == NERFHIP_BF16) {
                    auto load_frag = [&](const char* pb, int q) {
                        union { s16x4 h2[2]; bf16x8 v; } f;
                        f.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + tr_off + q * 512 + tr_s0));
                        f.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + tr_off + q * 512 + tr_s1));
                        return f.v;
                    };
                    // The tile's 2 x NXT MFMAs (two 16-point k-steps q, X tiles x) as ONE software pipeline pinned with sched_barriers
                    // (see mlp_bwd_dw_f8_kernel): the transposing reads of step m + RD - 1 are in flight when MFMA m issues, across
                    // the k-step boundary too (round 4: the pipeline used to drain and refill at every k-step — two exposed LDS
                    // round trips per ring stage with both waves of a SIMD in lock-step), and the bias sums (16 VALU per k-step)
                    // sit behind the first MFMAs instead of in front of them.
                    constexpr int RD = NERFHIP_DW_RD, NM = 2 * NXT;
                    const bf16x8 a0 = load_frag(dy_base, 0);
                    bf16x8 b[RD];
#pragma unroll
                    for (int m = 0; m < RD - 1; ++m)
                        if (m < NM) b[m] = load_frag(x_base + 2 * (m % NXT) * SLAB_BYTES, m / NXT);
                    const bf16x8 a1 = load_frag(dy_base, 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const int x = m % NXT;
                        if (m + RD - 1 < NM) b[(m + RD - 1) % RD] = load_frag(x_base + 2 * ((m + RD - 1) % NXT) * SLAB_BYTES, (m + RD - 1) / NXT);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m < NXT ? a0 : a1, b[m % RD], acc[x], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (m == 1) dw_bias_sum(a0, dbacc, dbacc2);
                        if (m == NXT + 1 || (NXT == 1 && m == 1)) dw_bias_sum(a1, dbacc, dbacc2);
                        if constexpr (SPREAD) {
                            if (m % DMA_STEP == DMA_STEP - 1 && m / DMA_STEP < LPW) {
                                issue_piece(m / DMA_STEP);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    if constexpr (SPREAD) {                            // (pieces the MFMA count did not reach)
#pragma unroll
                        for (int i = (2 * NXT) / DMA_STEP; i < LPW; ++i) issue_piece(i);
                    }
                } else {
#pragma unroll 4
                    for (int ks = 0; ks < 16; ++ks) {                  // 2 points per k-step
                        const int pt = 2 * ks + kk;
                        const float a = *reinterpret_cast<const float*>(dy_base + f32_off + pt * 32);
                        dbacc += a;
#pragma unroll
                        for (int x = 0; x < NXT; ++x) {
                            const float b = *reinterpret_cast<const float*>(x_base + 2 * x * SLAB_BYTES + f32_off + pt * 32);
                            acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[x], 0, 0, 0);
                        }
                    }
                }
            } else if constexpr (FOLD) {
                // the folded sigma head: wave w = 4..7, dY_sigma (one tile) x the h8 tiles 2 (w - 4), 2 (w - 4) + 1 = X tiles 1 + .. of
                // the stage (tile 0 is enc_dir) into acc[0], acc[1]; wave 4 also sums dY_sigma for the bias
                const char* sg = st_base + kDwFoldSigmaSlab * SLAB_BYTES;
                const char* xs = st_base + (jb.dy_slabs + 2 * (1 + 2 * (wave - kDwFoldRow0))) * SLAB_BYTES;
                if constexpr (PREC == NERFHIP_BF16) {
                    auto load_frag = [&](const char* pb, int q) {
                        union { s16x4 h2[2]; bf16x8 v; } f;
                        f.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + tr_off + q * 512 + tr_s0));
                        f.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + tr_off + q * 512 + tr_s1));
                        return f.v;
                    };
                    const bf16x8 a0 = load_frag(sg, 0), b00 = load_frag(xs, 0), b10 = load_frag(xs + 2 * SLAB_BYTES, 0);
                    const bf16x8 a1 = load_frag(sg, 1), b01 = load_frag(xs, 1), b11 = load_frag(xs + 2 * SLAB_BYTES, 1);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b00, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b10, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b01, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b11, acc[1], 0, 0, 0);
                    if (wave == kDwFoldRow0) {
                        dw_bias_sum(a0, dbacc, dbacc2);
                        dw_bias_sum(a1, dbacc, dbacc2);
                    }
                } else {
#pragma unroll 4
                    for (int ks = 0; ks < 16; ++ks) {
                        const int pt = 2 * ks + kk;
                        const float a = *reinterpret_cast<const float*>(sg + f32_off + pt * 32);
                        dbacc += a;
                        const float b0 = *reinterpret_cast<const float*>(xs + f32_off + pt * 32);
                        const float b1 = *reinterpret_cast<const float*>(xs + 2 * SLAB_BYTES + f32_off + pt * 32);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
                    }
                }
            }
            // (REGEN) the NEXT stage's encoding slabs, behind this stage's MFMAs: its depths came with stage `it`'s pieces (landed at
            // this iteration's wait); the writes are visible to all behind the next barrier (whose wait includes lgkmcnt(0))
            if constexpr (REGEN) gen_stage(it + 1);
#if NERFHIP_DW_PROBE
            pr_comp += (shader_cycles() - t3) & 0xffffffffu;
#endif
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the look-ahead DMAs before exit
        // this workgroup's partial sums: [dY tile][X tile] blocks of 1024 floats (dw_store_block) + one bias row per dY tile
        float* sl = slabs + (size_t)blockIdx.x * kDwSlabFloats;
        dbacc += dbacc2;
        if constexpr (SPLIT2D) {
            constexpr int XW = NXT / 2;
            const int wi = wave >> 1, wj = wave & 1;
#pragma unroll
            for (int xl = 0; xl < XW; ++xl)
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    dw_store_block(sl + (size_t)((2 * wi + d) * kDwMaxXTiles + XW * wj + xl) * 1024, acc[2 * xl + d], lane);
            sl[8 * kDwMaxXTiles * 64 * 16 + (2 * wi + wj) * 64 + lane] = dbacc;
        } else if (wave < n_ot) {
#pragma unroll
            for (int x = 0; x < NXT; ++x) dw_store_block(sl + (size_t)(wave * kDwMaxXTiles + x) * 1024, acc[x], lane);
            sl[8 * kDwMaxXTiles * 64 * 16 + wave * 64 + lane] = dbacc;
        } else if constexpr (FOLD) {             // the sigma head's partials: blocks dw_fold_block(2 (w - 4)), (.. + 1); bias row kDwFoldRow0
            dw_store_block(sl + (size_t)dw_fold_block(2 * (wave - kDwFoldRow0)) * 1024, acc[0], lane);
            dw_store_block(sl + (size_t)dw_fold_block(2 * (wave - kDwFoldRow0) + 1) * 1024, acc[1], lane);
            if (wave == kDwFoldRow0) sl[8 * kDwMaxXTiles * 64 * 16 + kDwFoldRow0 * 64 + lane] = dbacc;
        }
    };
    // job classes of mlp_layout.h kDwJobs: (X tiles, dY + X slabs per stage)
    using std::integral_constant;
    using std::false_type;
    using std::true_type;
    bool regen = false;
    if constexpr (PREC == NERFHIP_BF16) regen = enc_rays != nullptr;         // (host: only with the sigma head folded into the dir job)
    switch (n_xt) {
        case 2:                                                                                        // first layer: 16 + 4
            if constexpr (PREC == NERFHIP_BF16) {
                if (regen) { run(integral_constant<int, 2>{}, integral_constant<int, 20>{}, true_type{}); break; }
            }
            run(integral_constant<int, 2>{}, integral_constant<int, 20>{}, false_type{});
            break;
        case 4: run(integral_constant<int, 4>{}, integral_constant<int, 10>{}, false_type{}); break;   // rgb head: 2 + 8
        case 8:
            if (jb.dy_slabs == 16) run(integral_constant<int, 8>{}, integral_constant<int, 32>{}, false_type{});     // 256 x 256 layers: 16 + 16
            else run(integral_constant<int, 8>{}, integral_constant<int, 18>{}, false_type{});         // sigma head on its own: 2 + 16
            break;
        case 9:
            if (jobs.fold_of[(jid / kNumDwJobs) * kNumDwJobs + kDwJobSigma] == jid) {                  // dir layer + folded sigma head: 8 + 18 + 2
                if constexpr (PREC == NERFHIP_BF16) {
                    if (regen) { run(integral_constant<int, 9>{}, integral_constant<int, kDwFoldStageSlabs>{}, true_type{}); break; }
                }
                run(integral_constant<int, 9>{}, integral_constant<int, kDwFoldStageSlabs>{}, false_type{});
            } else {
                run(integral_constant<int, 9>{}, integral_constant<int, 26>{}, false_type{});          // dir layer: 8 + 18
            }
            break;
        default:                                                                                       // skip layer: 16 + 20 (kDwMaxXTiles)
            if constexpr (PREC == NERFHIP_BF16) {
                if (regen) { run(integral_constant<int, 10>{}, integral_constant<int, 36>{}, true_type{}); break; }
            }
            run(integral_constant<int, 10>{}, integral_constant<int, 36>{}, false_type{});
            break;
    }
#if NERFHIP_DW_PROBE
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned* pr = g_dw_probe + ((size_t)blockIdx.x * 8 + wave) * 8;
        pr[0] = (unsigned)my_tiles; pr[1] = pr_wait; pr[2] = pr_bar; pr[3] = pr_issue; pr[4] = pr_comp;
        pr[5] = (unsigned)(__builtin_amdgcn_s_memrealtime() - pr_t00);        // 100 MHz ticks
        pr[6] = (unsigned)jid; pr[7] = pr_depth;
    }
#endif
}

// every job of mlp_layout.h has one of the kernel's classes (the switch above)
NH_HD constexpr bool dw_job_has_class(const DwJob& j) {
    const int nxt = (j.x1_slabs + j.x2_slabs) / 2, nsl = j.dy_slabs + j.x1_slabs + j.x2_slabs;
    return (nxt == 2 && nsl == 20) || (nxt == 4 && nsl == 10) || (nxt == 8 && ((nsl == 32 && j.dy_slabs == 16) || (nsl == 18 && j.dy_slabs != 16))) ||
           (nxt == 9 && nsl == 26) || (nxt == 10 && nsl == 36);
}
NH_HD constexpr bool dw_jobs_have_classes() {
    for (int j = 0; j < kNumDwJobs; ++j)
        if (!dw_job_has_class(kDwJobs[j])) return false;
    return true;
}
static_assert(dw_jobs_have_classes(), "mlp_bwd_dw_kernel: a weight-gradient job without a compiled job class");

// ================================================================================================
// Phase B, fp8 storage (NERFHIP_BF16_F8): dW = dY^T X on v_mfma_scale_f32_32x32x64_f8f6f4
// ================================================================================================
// Same decomposition as mlp_bwd_dw_kernel (workgroup = (layer job, point split), wave w = 32 dY features x all X tiles,
// fp32 accumulators in registers), but the operands are the e4m3 slab-pair pieces of mlp_layout.h ("fp8 storage"): one
// 1 KiB piece = 32 points x 32 features, i.e. HALF the bytes per point of the bf16 kernel, and one MFMA consumes K = 64
// points = two wave tiles per iteration.
//   operand fragment of v_mfma_scale_f32_32x32x64_f8f6f4 (measured, tools/probes/probe_fp8.hip): lane (row m = l & 31,
//   H = l >> 5) holds 32 bytes; bytes 0..15 belong to K block 0, bytes 16..31 to K block 1 (for both lane halves); the
//   scale operand of lanes 0..31 scales block 0 of row m, that of lanes 32..63 block 1.
//   => block 0 = tile T0, block 1 = tile T1 of the iteration; lane half H supplies points 16H .. 16H+15 of each.
//   ds_read_b64_tr_b8 (measured): within a 16-lane group, result lane c (column c & 7, row parity c >> 3) byte b = byte
//   (c & 7) of the 8-byte chunk addressed by source lane 2b + (c >> 3).  Source lane r therefore points at the chunk of
//   (point 8g + (r >> 1), half r & 1) and the group's 16 result lanes become the 16 features (h = c >> 3, j = c & 7) of one
//   slab with 8 consecutive points in their bytes.
// LDS image of a piece: 16-byte unit u = 16 g + 8 h + (n & 7) <- global unit (lane) 32 h + n, n = 8 g + (n & 7): the 32
// lanes of one ds_read pass (2 slabs x 8 points x 2 halves) cover one aligned 256-byte block => conflict free.
struct DwF8Job {
    int dy_pair0, dy_pairs, dy_pos0;               // pieces / scale-table index of the dY section
    int x1_pair0, x1_pairs, x1_pos0;
    int x2_pair0, x2_pairs, x2_pos0;
};

__device__ __forceinline__ void glds4b(const void* gsrc, unsigned lds_dst) {      // 4 bytes per lane, global -> LDS
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;

#ifndef NERFHIP_DWF8_DEPTH
#define NERFHIP_DWF8_DEPTH 4
#endif

__global__ __launch_bounds__(512, 2)
void mlp_bwd_dw_f8_kernel(DwJobTable jobs, float* __restrict__ slabs) {
    constexpr int DEPTH = NERFHIP_DWF8_DEPTH;
    constexpr int MAXP = 36;                                   // pieces per stage: 2 tiles x (8 dY + 10 X) pairs
    constexpr int LPW = 5;                                     // piece DMAs per wave per stage (8 x 5 >= 36)
    constexpr int STAGE_BYTES = MAXP * kPieceBytes;
    constexpr int SCALE_BYTES = 256;                           // per wave per stage: [tile][16 dwords]
    __shared__ __attribute__((aligned(1024))) char ring[DEPTH * STAGE_BYTES + DEPTH * 8 * SCALE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int jid = 0;
#pragma unroll
    for (int j = 1; j < kDwMaxJobs; ++j) jid += ((int)blockIdx.x >= jobs.soff[j]) ? 1 : 0;
    const int nsplit = jobs.nsplit[jid], split = (int)blockIdx.x - jobs.soff[jid];
    const DwJob jb = jobs.job[jid];
    const int64_t ntiles = jobs.ntiles[jid];
    const uint8_t* __restrict__ acts_base = jobs.acts[jid];
    const uint8_t* __restrict__ dys_base = jobs.dys[jid];
    const int dyp = jb.dy_slabs / 2, x1p = jb.x1_slabs / 2, x2p = jb.x2_slabs / 2;
    const int n_ot = dyp, n_xt = x1p + x2p;
    // the dir layer's workgroups also form the sigma head's gradient (same X section h8, read once; see mlp_bwd_dw_kernel): their tile
    // carries the dY_sigma pair as its LAST piece and the section's scale in slot 3 of the wave's scale dwords; the waves 4..7 (the
    // dir layer has 4 dY tiles) multiply it by the h8 tiles 2 (w - 4), 2 (w - 4) + 1
    const bool fold = jobs.fold_of[(jid / kNumDwJobs) * kNumDwJobs + kDwJobSigma] == jid;
    const int np = dyp + n_xt + (fold ? 1 : 0);                // pieces per tile
    const int dy_pair0 = jb.dy_off / 2, x1_pair0 = jb.x1_off / 2, x2_pair0 = jb.x2_off / 2;
    // tile PAIRS per split (K = 64 points per MFMA); ntiles is a multiple of 8
    const int64_t npairs = ntiles / 2;
    const int64_t per = (npairs + nsplit - 1) / nsplit;
    const int64_t p_first = (int64_t)split * per;
    const int64_t my_pairs = (p_first >= npairs) ? 0 : ((npairs - p_first < per) ? npairs - p_first : per);
    const unsigned lds_base = (unsigned)(uintptr_t)ring;
    const unsigned lds_scales = lds_base + (unsigned)(DEPTH * STAGE_BYTES);

    // DMA source unit of LDS unit `lane` (see the header comment): global lane 32 h + 8 g + (n & 7)
    const int dma_unit = ((lane >> 3) & 1) * 32 + 8 * (lane >> 4) + (lane & 7);
    // scale DMA: lane i < 32 -> (tile i >> 4, slot i & 15) lands at dword i of the wave's scale area: slot 0 = the dY
    // section's scale, slot 1 = the x1 section's, slots >= 2 = the x2 section's (x1's when there is no x2)
    const int s_tile = (lane >> 4) & 1, s_slot = lane & 15;
    const bool s_sigma = fold && s_slot == 3;
    const int s_from_dy = s_slot == 0 || s_sigma;
    const int s_pos = s_sigma ? f8_dy_section(kDySigma)
                              : (s_slot == 0 ? f8_dy_section(jb.dy_off)
                                             : ((s_slot == 1 || x2p == 0) ? f8_x_section(jb.x1_off) : f8_x_section(jb.x2_off)));
    // which piece of a tile pair each of this wave's LPW DMAs fetches does not depend on the stage: (dY or X block, byte offset from
    // the pair's first tile block, LDS offset in the stage) once, ahead of the loop (wave-uniform; the stage loop only adds the pair's
    // two block pointers — the selection used to be a branch ladder per DMA)
    bool p_dy[LPW];
    unsigned p_off[LPW], p_dst[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        int pi = wave + 8 * i;
        if (pi >= 2 * np) pi = 2 * np - 1;                                           // duplicate DMA of the last piece
        const int tl = pi >= np ? 1 : 0, pp = pi - tl * np;
        int pair;
        if (pp < dyp) { p_dy[i] = true; pair = dy_pair0 + pp; }
        else if (fold && pp == np - 1) { p_dy[i] = true; pair = kDySigma / 2; }
        else if (pp < dyp + x1p) { p_dy[i] = false; pair = x1_pair0 + pp - dyp; }
        else { p_dy[i] = false; pair = x2_pair0 + pp - dyp - x1p; }
        p_off[i] = (unsigned)(pair * kPieceBytes + tl * (p_dy[i] ? f8_dy_tile_bytes() : f8_act_tile_bytes()));
        p_dst[i] = (unsigned)(pi * kPieceBytes);
    }
    auto issue_stage = [&](int64_t it) {
        int64_t P = p_first + (it < my_pairs ? it : my_pairs - 1);                   // past the end: re-fetch the last pair
        if (P >= npairs) P = npairs - 1;
        const unsigned slot = lds_base + (unsigned)((it % DEPTH) * STAGE_BYTES);
        const uint8_t* dyb = dys_base + (size_t)(2 * P) * f8_dy_tile_bytes();
        const uint8_t* acb = acts_base + (size_t)(2 * P) * f8_act_tile_bytes();
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            glds16b_nt((p_dy[i] ? dyb : acb) + p_off[i] + dma_unit * 16, slot + p_dst[i]);
        {
            const uint8_t* src = s_from_dy ? dyb + (size_t)s_tile * f8_dy_tile_bytes() + f8_dy_scale_off()
                                           : acb + (size_t)s_tile * f8_act_tile_bytes() + f8_act_scale_off();
            glds4b(src + 4 * s_pos, lds_scales + (unsigned)(((it % DEPTH) * 8 + wave) * SCALE_BYTES));
        }
    };

    f32x16 acc[kDwMaxXTiles];
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.0f;
#pragma unroll
    for (int x = 0; x < kDwMaxXTiles; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;

#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) issue_stage(s);
#if NERFHIP_DW_PROBE
    unsigned pr_wait = 0, pr_bar = 0, pr_issue = 0, pr_comp = 0;
    const uint64_t pr_t00 = __builtin_amdgcn_s_memrealtime();
#endif

    // per-lane read geometry: H = lane >> 5 (points 16H..16H+15 of each tile), s = slab of the pair, r = source row
    const int H = lane >> 5, sl = (lane >> 4) & 1, r = lane & 15;
    const int rd_off = ((2 * H) * 16 + (r & 1) * 8 + (r >> 1)) * 16 + sl * 8;        // read q: + (q & 1) * 256, tile (q >> 1): + np KiB
    const int tile1 = np * kPieceBytes;
    i32x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = 0x38383838;                                 // e4m3 1.0

    // The iteration loop exists once per X-tile count of the jobs (2 first layer, 4 rgb head, 8 the 256 x 256 layers and the sigma
    // head, 9 dir layer, 10 skip layer), chosen by ONE wave-uniform switch outside it: with n_xt a compile-time constant the X
    // loop is straight-line code, the next tile's four transposing LDS reads are in flight while the current tile's MFMA issues,
    // and the compiler schedules across tiles.  (With the runtime guard `if (x < n_xt)` every tile was a branch target of its own:
    // 4 ds_read -> s_waitcnt lgkmcnt(0) -> MFMA, ten times per iteration in the same registers — the kernel was bound by ten
    // exposed LDS round trips per ring stage, not by HBM: "a workgroup's time follows its iteration count, not its bytes".)
    auto run = [&](auto nxt_c, auto fold_c) {
        constexpr int NXT = decltype(nxt_c)::value;
        constexpr bool FOLD = decltype(fold_c)::value;
        for (int64_t it = 0; it < my_pairs; ++it) {
            // stage `it` landed (DEPTH-2 younger stages of LPW + 1 DMAs may still fly), everyone done with stage it-1
#if NERFHIP_DW_PROBE
            const unsigned t0 = shader_cycles();
            wait_vm<(DEPTH - 2) * (LPW + 1)>();
            const unsigned t1 = shader_cycles();
            asm volatile("s_barrier" ::: "memory");
            const unsigned t2 = shader_cycles();
            pr_wait += t1 - t0;
            pr_bar += t2 - t1;
#else
            wait_vm_barrier<(DEPTH - 2) * (LPW + 1)>();
#endif
            issue_stage(it + DEPTH - 1);
#if NERFHIP_DW_PROBE
            const unsigned t3 = shader_cycles();
            pr_issue += t3 - t2;
#endif
            if (wave < n_ot) {
                const char* st_base = ring + (it % DEPTH) * STAGE_BYTES + rd_off;
                const char* sc_base = ring + DEPTH * STAGE_BYTES + ((it % DEPTH) * 8 + wave) * SCALE_BYTES + H * 64;
                auto load_frag = [&](const char* pb) {
                    i32x8 f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
                            (__attribute__((address_space(3))) i32x2*)(pb + (q >> 1) * tile1 + (q & 1) * 256));
                        f[2 * q] = v[0];
                        f[2 * q + 1] = v[1];
                    }
                    return f;
                };
                const char* x_base = st_base + dyp * kPieceBytes;
                // software pipeline, pinned with sched_barriers (left alone, hipcc sinks every tile's reads back to just before
                // its MFMA: one exposed LDS round trip per tile): the reads of tiles x + 1 and x + 2 are in flight when MFMA x issues
                constexpr int RD = 3;
                const i32x8 a = load_frag(st_base + wave * kPieceBytes);
                const int sa = *reinterpret_cast<const int*>(sc_base);
                const int sx1 = *reinterpret_cast<const int*>(sc_base + 4), sx2 = *reinterpret_cast<const int*>(sc_base + 8);
                i32x8 b[RD];
                b[0] = load_frag(x_base);
                if (NXT > 1) b[1] = load_frag(x_base + kPieceBytes);
                __builtin_amdgcn_sched_barrier(0);
                accb = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, ones, accb, NERFHIP_F8_DY_E5M2, 0, 0, sa, 0, 127);   // bias: dY x 1.0
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int x = 0; x < NXT; ++x) {
                    if (x + 2 < NXT) b[(x + 2) % RD] = load_frag(x_base + (x + 2) * kPieceBytes);
                    __builtin_amdgcn_sched_barrier(0);
                    // A = dY: e5m2 (cbsz 1), B = X: e4m3 (blgp 0); lanes 0..31 carry tile T0's section scales, lanes 32..63 T1's
                    acc[x] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b[x % RD], acc[x], NERFHIP_F8_DY_E5M2, 0, 0, sa, 0, x < x1p ? sx1 : sx2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (FOLD) {
                // the folded sigma head: dY_sigma pair (the tile's last piece) x X pieces 1 + 2 (w - 4), 2 + 2 (w - 4) of the stage (piece
                // 0 is enc_dir) into acc[0], acc[1]; wave 4 also forms the bias partial (dY_sigma x 1.0)
                const char* st_base = ring + (it % DEPTH) * STAGE_BYTES + rd_off;
                const char* sc_base = ring + DEPTH * STAGE_BYTES + ((it % DEPTH) * 8 + wave) * SCALE_BYTES + H * 64;
                auto load_frag = [&](const char* pb) {
                    i32x8 f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
                            (__attribute__((address_space(3))) i32x2*)(pb + (q >> 1) * tile1 + (q & 1) * 256));
                        f[2 * q] = v[0];
                        f[2 * q + 1] = v[1];
                    }
                    return f;
                };
                const char* xs = st_base + (dyp + 1 + 2 * (wave - kDwFoldRow0)) * kPieceBytes;
                const i32x8 a_sg = load_frag(st_base + (np - 1) * kPieceBytes);
                const i32x8 b0 = load_frag(xs), b1 = load_frag(xs + kPieceBytes);
                const int sa_sg = *reinterpret_cast<const int*>(sc_base + 12), sx2 = *reinterpret_cast<const int*>(sc_base + 8);
                acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_sg, b0, acc[0], NERFHIP_F8_DY_E5M2, 0, 0, sa_sg, 0, sx2);
                acc[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_sg, b1, acc[1], NERFHIP_F8_DY_E5M2, 0, 0, sa_sg, 0, sx2);
                if (wave == kDwFoldRow0)
                    accb = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_sg, ones, accb, NERFHIP_F8_DY_E5M2, 0, 0, sa_sg, 0, 127);
            }
#if NERFHIP_DW_PROBE
            pr_comp += shader_cycles() - t3;
#endif
        }
    };
    switch (n_xt) {
        case 2: run(std::integral_constant<int, 2>{}, std::false_type{}); break;
        case 4: run(std::integral_constant<int, 4>{}, std::false_type{}); break;
        case 8: run(std::integral_constant<int, 8>{}, std::false_type{}); break;
        case 9:
            if (fold) run(std::integral_constant<int, 9>{}, std::true_type{});       // dir layer + folded sigma head
            else run(std::integral_constant<int, 9>{}, std::false_type{});
            break;
        default: run(std::integral_constant<int, 10>{}, std::false_type{}); break;          // 10 = kDwMaxXTiles (the skip layer)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // drain the look-ahead DMAs before exit

    if (wave < n_ot) {
        float* slb = slabs + (size_t)blockIdx.x * kDwSlabFloats;
#pragma unroll
        for (int x = 0; x < kDwMaxXTiles; ++x) {
            if (x < n_xt) dw_store_block(slb + (size_t)(wave * kDwMaxXTiles + x) * 1024, acc[x], lane);
        }
        // bias partials: every column of accb equals sum_p dY[p][row]; lanes 0 and 32 hold column 0 (rows 4H + (r&3) + 8(r>>2))
        if ((lane & 31) == 0) {
            float* bdst = slb + 8 * kDwMaxXTiles * 64 * 16 + wave * 64;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) bdst[(rr & 3) + 8 * (rr >> 2) + 4 * H] = accb[rr];
        }
    } else if (fold) {                           // the sigma head's partials: blocks dw_fold_block(2 (w - 4)), (.. + 1); bias row kDwFoldRow0
        float* slb = slabs + (size_t)blockIdx.x * kDwSlabFloats;
        dw_store_block(slb + (size_t)dw_fold_block(2 * (wave - kDwFoldRow0)) * 1024, acc[0], lane);
        dw_store_block(slb + (size_t)dw_fold_block(2 * (wave - kDwFoldRow0) + 1) * 1024, acc[1], lane);
        if (wave == kDwFoldRow0 && (lane & 31) == 0) {
            float* bdst = slb + 8 * kDwMaxXTiles * 64 * 16 + kDwFoldRow0 * 64;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) bdst[(rr & 3) + 8 * (rr >> 2) + 4 * H] = accb[rr];
        }
    }
#if NERFHIP_DW_PROBE
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned* pr = g_dw_probe + ((size_t)blockIdx.x * 8 + wave) * 8;
        pr[0] = (unsigned)my_pairs; pr[1] = pr_wait; pr[2] = pr_bar; pr[3] = pr_issue; pr[4] = pr_comp;
        pr[5] = (unsigned)(__builtin_amdgcn_s_memrealtime() - pr_t00);        // 100 MHz ticks
        pr[6] = (unsigned)jid; pr[7] = DEPTH;
    }
#endif
}


struct GradTable {
    float* w[kDwMaxJobs];     // per JOB: job j writes parameter tensor kDwJobs[j % 12].param of model j / 12
    float* b[kDwMaxJobs];
};

// Adam fused into the reduce (single-GPU training step: no all-reduce sits between the gradients and the update).  A model's
// parameters, exp_avg and exp_avg_sq live in flat fp32 buffers laid out exactly like its flat gradient buffer (optim.py
// FlatAdam, ops.mlp_bwd), so gradient element e updates element e of each.  `state` = {step count, arrival ticket} as in
// adam_kernel (optim.hip); nullptr = plain reduce.
struct AdamFused {
    float* param[kDwMaxModels];
    float* m[kDwMaxModels];
    float* v[kDwMaxModels];
    const float* grad0[kDwMaxModels];     // base of the model's flat gradient buffer
    float* state;
    float lr, beta1, beta2, eps, wd;
};

// one gradient element: written (or accumulated) and, with Adam fused, applied
struct GradEmit {
    const AdamFused& A;
    AdamCoef ac;
    int accumulate, model;
    __device__ __forceinline__ void operator()(float* dst, float val) const {
        const float g = accumulate ? *dst + val : val;
        *dst = g;
        if (A.state) {
            const size_t e = (size_t)(dst - A.grad0[model]);
            adam_elem(A.param[model][e], g, A.m[model][e], A.v[model][e], ac, A.beta2, A.eps, A.wd);
        }
    }
};

// sum split slabs, undo the fragment/feature permutation, write (out,in) row-major gradients [and apply Adam].
// Columns of enc kDwEncFold (the dir job's h8 section) are the G matrix of the folded final layer, the dir job's bias sums its s:
// both also go — plainly — to the model's fold scratch for mlp_bwd_fold_kernel; the final layer's own job has nothing here.
// One 256-thread block per (job, 32x32 tile): thread = one float4 (rows o..o+3 of one column) of the 1024-float
// tile, summed over the job's splits with independent 16-B loads.
// F8: operand rows/columns arrive in the order ds_read_b64_tr_b8 delivers them (m -> slab m >> 4, half (m >> 3) & 1, slot m & 7)
// instead of natural feature order, and the bias partials hold one value per row.
template <bool F8>
__global__ __launch_bounds__(256) void mlp_bwd_reduce_kernel(DwJobTable jobs, const float* __restrict__ slabs, GradTable G,
                                                              float* __restrict__ fold_scratch, int accumulate, AdamFused A) {
    const int jid = blockIdx.y;
    const DwJob jb = jobs.job[jid];
    const int model = jid / kNumDwJobs;
    const int fold = jobs.fold_of[jid];                 // >= 0: this job's partials live in job `fold`'s slabs (blocks dw_fold_block(X tile))
    const int nsplit = jobs.nsplit[fold >= 0 ? fold : jid], s0 = jobs.soff[fold >= 0 ? fold : jid];
    const bool derived = jid % kNumDwJobs == kDwJobFinal;              // finished by mlp_bwd_fold_kernel
    const int n_ot = derived ? 0 : jb.dy_slabs / 2, n_xt = (jb.x1_slabs + jb.x2_slabs) / 2;
    float* const scratch = fold_scratch + (size_t)model * kFoldScratchFloats;
    const int n_out = kParamOut[jb.param], ldw = kParamIn[jb.param];
    const int tile = blockIdx.x;                       // (ot, xt) pairs + one extra block per ot for the bias
    const int ot = tile / (kDwMaxXTiles + 1), xt = tile % (kDwMaxXTiles + 1);
    AdamCoef ac;
    if (A.state) ac = adam_coef(A.state[0] + 1.0f, A.lr, A.beta1, A.beta2);
    const GradEmit emit{A, ac, accumulate, model};
    if (ot < n_ot && xt == kDwMaxXTiles) {             // bias: lanes (m,0) + (m,1)
        const int m = threadIdx.x;
        if (m < 32) {
            float sacc = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) {
                const float* sl = slabs + (size_t)(s0 + sp) * kDwSlabFloats + (size_t)8 * kDwMaxXTiles * 64 * 16 +
                                  (fold >= 0 ? kDwFoldRow0 : ot) * 64;
                sacc += F8 ? sl[m] : sl[m] + sl[m + 32];
            }
            const int o = F8 ? 32 * ot + chain_feature(m >> 4, f8_row_h(m & 15), f8_row_j(m & 15)) : 32 * ot + m;
            if (o < n_out) {
                emit(G.b[jid] + o, sacc);
                if (jid % kNumDwJobs == kDwJobDir) scratch[kFoldS + o] = sacc;
            }
        }
    } else if (ot < n_ot && xt < n_xt) {
        const int e4 = threadIdx.x;                    // float4 e4 of the block (dw_store_block: register-major): lane = e4 & 63, r = 4*(e4>>6)+k
        const float4* src = reinterpret_cast<const float4*>(slabs + (size_t)s0 * kDwSlabFloats +
                                                            ((size_t)(fold >= 0 ? dw_fold_block(xt) : ot * kDwMaxXTiles + xt) * 64) * 16) + e4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int sp = 0; sp < nsplit; ++sp) {
            const float4 v = src[(size_t)sp * (kDwSlabFloats / 4)];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const int lane = e4 & 63, rq = e4 >> 6;
        const int h = lane >> 5, ncol = lane & 31;
        const int m0 = 8 * rq + 4 * h;                     // operand rows m0 .. m0+3  (reg r = 4*rq + k -> (r&3) = k, r>>2 = rq)
        // natural order: row m = feature 32 ot + m.  F8: m -> feature chain_feature(m >> 4, (m >> 3) & 1, m & 7) of the tile
        const int o0 = F8 ? 32 * ot + chain_feature(m0 >> 4, f8_row_h(m0 & 15), f8_row_j(m0 & 15)) : 32 * ot + m0;   // k adds to (m & 3)
        const int xi = 32 * xt + ncol;
        int xs = xi >> 4;
        const int i = xi & 15;
        const int sh = F8 ? f8_row_h(i) : slab_nat_h(i), sj = F8 ? f8_row_j(i) : slab_nat_j(i);     // slot (h, j) inside slab xs
        int enc, col0;
        if (xs < jb.x1_slabs) { enc = jb.x1_enc; col0 = jb.x1_col0; }
        else { xs -= jb.x1_slabs; enc = jb.x2_enc; col0 = jb.x2_col0; }
        int col;
        if (enc == 0 || enc == kDwEncFold) col = col0 + chain_feature(xs, sh, sj);
        else {
            const int ch = (enc == 1) ? xyz_slot_channel(xs, sh, sj) : dir_slot_channel(xs, sh, sj);
            col = ch < 0 ? -1 : col0 + ch;
        }
        if (enc == kDwEncFold) {                           // G[o][h8 feature]: this call's sum, never accumulated, never an Adam input
            const float vals[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (o0 + k < n_out) scratch[kFoldG + (size_t)(o0 + k) * kW + col] = vals[k];
        } else if (col >= 0 && col < ldw) {
            const float vals[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = o0 + k;
                if (o < n_out) emit(G.w[jid] + (size_t)o * ldw + col, vals[k]);
            }
        }
    }
    if (A.state) {
        // arrival ticket (as adam_kernel): the last workgroup of the launch advances the step counter, after every workgroup
        // that uses it has read the old value
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* ticket = reinterpret_cast<unsigned*>(A.state + 1);
            const unsigned prev = atomicAdd(ticket, 1u);
            if (prev == gridDim.x * gridDim.y - 1) {
                *ticket = 0u;
                A.state[0] = A.state[0] + 1.0f;
            }
        }
    }
}

// ================================================================================================
// Phase C: the folded final layer (mlp_layout.h kDwJobs)
// ================================================================================================
//     dW_dir[j][m]   = sum_k G[j][k] W_f[m][k] + s[j] b_f[m]      (j < 128, m < 256: the first 256 columns of dir_encoding's weight)
//     dW_final[m][k] = sum_j W_dx[j][m] G[j][k]                    (m, k < 256)
//     db_final[m]    = sum_j W_dx[j][m] s[j]
// fp32 FMAs in a fixed order (2 x 8.4 M per model).  One 256-thread workgroup per 32 x 32 output tile, operands staged through LDS
// in 32-deep slices: blocks 0..31 the dW_dir tiles, 32..95 the dW_final tiles, 96 the bias.  W_f, W_dx, b_f come from the fold
// block of the packed W^T image — a snapshot taken before the step's update, so with Adam fused the update of one block cannot
// reach the operands of another.
struct FoldArgs {
    const float* image[kDwMaxModels];      // fold block of the model's packed W^T image
    float* gw_final[kDwMaxModels];
    float* gb_final[kDwMaxModels];
    float* gw_dir[kDwMaxModels];
};
constexpr int kFoldBlocks = 32 + 64 + 1;
// Latency, not arithmetic, is what this launch costs (it sits between the reduce and the optimizer): a workgroup fetches BOTH
// operands of its tile whole — one round trip to L2 / HBM — and only then multiplies out of LDS, on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: an fmaf chain per output, as in the fp32 kernels), each of its 4 waves over a quarter of the inner
// dimension; the four partial tiles are summed in a fixed order.  (First version: 32-deep slices, eight dependent round trips, 15 us
// in the step; second: one round trip and a VALU loop bound by its LDS reads, ~6 us.)
__global__ __launch_bounds__(256) void mlp_bwd_fold_kernel(FoldArgs F, const float* __restrict__ fold_scratch, int accumulate, AdamFused A) {
    constexpr int PA = 257, PB = 33;                      // LDS row pitches (floats): conflict-free reads
    __shared__ float lds[2 * 32 * PA];
    __shared__ float part[4][16][64];
    const int model = blockIdx.y, bx = blockIdx.x, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const float* __restrict__ Gm = fold_scratch + (size_t)model * kFoldScratchFloats + kFoldG;
    const float* __restrict__ sv = fold_scratch + (size_t)model * kFoldScratchFloats + kFoldS;
    const float* __restrict__ Wf = F.image[model] + (size_t)kFoldWf * 256;
    const float* __restrict__ Wdx = F.image[model] + (size_t)kFoldWdx * 256;
    const float* __restrict__ bf = F.image[model] + (size_t)kFoldBf * 256;
    AdamCoef ac;
    if (A.state) ac = adam_coef(A.state[0], A.lr, A.beta1, A.beta2);          // (the reduce launch before this one advanced the counter)
    const GradEmit emit{A, ac, accumulate, model};
    if (bx == kFoldBlocks - 1) {
        // db_final[t] = sum_j W_dx[j][t] s[j]: four partial chains, the loads independent
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int j = 0; j < 128; j += 4)
#pragma unroll
            for (int q = 0; q < 4; ++q) a4[q] = __builtin_fmaf(Wdx[(size_t)(j + q) * 256 + t], sv[j + q], a4[q]);
        emit(F.gb_final[model] + t, (a4[0] + a4[1]) + (a4[2] + a4[3]));
        return;
    }
    const bool dir = bx < 32;
    // tile origin: dW_dir rows j0.. x columns m0.. | dW_final rows m0.. x columns k0..
    const int row0 = dir ? 32 * (bx >> 3) : 32 * ((bx - 32) >> 3), col0 = dir ? 32 * (bx & 7) : 32 * ((bx - 32) & 7);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int m = lane & 31, kh = lane >> 5;              // MFMA operand lane: row / column m, inner index parity kh
    if (dir) {
        // out[j][c] = sum_k G[j0 + j][k] W_f[m0 + c][k]:  A[j][k] = G rows, B[k][c] = W_f rows; both staged row-major, pitch PA
        float* sa = lds;
        float* sb = lds + 32 * PA;
        float4 va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {                     // row (t >> 6) + 4 i, floats 4 (t & 63) ..
            const int row = (t >> 6) + 4 * i;
            va[i] = reinterpret_cast<const float4*>(Gm + (size_t)(row0 + row) * 256)[t & 63];
            vb[i] = reinterpret_cast<const float4*>(Wf + (size_t)(col0 + row) * 256)[t & 63];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = ((t >> 6) + 4 * i) * PA + 4 * (t & 63);
            sa[o] = va[i].x; sa[o + 1] = va[i].y; sa[o + 2] = va[i].z; sa[o + 3] = va[i].w;
            sb[o] = vb[i].x; sb[o + 1] = vb[i].y; sb[o + 2] = vb[i].z; sb[o + 3] = vb[i].w;
        }
        __syncthreads();
        const float* pa = sa + m * PA + 64 * wave + kh;   // this wave's quarter of k
        const float* pb = sb + m * PA + 64 * wave + kh;
#pragma unroll 8
        for (int k = 0; k < 64; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
    } else {
        // out[r][c] = sum_j W_dx[j][m0 + r] G[j][k0 + c]:  A[r][j] = W_dx columns, B[j][c] = G rows; staged [j][32], pitch PB
        float* sa = lds;
        float* sb = lds + 128 * PB;
        const int c = t & 31, r0 = t >> 5;
        float va[16], vb[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {                    // row r0 + 8 i (of 128), column c
            va[i] = Wdx[(size_t)(r0 + 8 * i) * 256 + row0 + c];
            vb[i] = Gm[(size_t)(r0 + 8 * i) * 256 + col0 + c];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sa[(r0 + 8 * i) * PB + c] = va[i];
            sb[(r0 + 8 * i) * PB + c] = vb[i];
        }
        __syncthreads();
        const float* pa = sa + (32 * wave + kh) * PB + m; // this wave's quarter of j
        const float* pb = sb + (32 * wave + kh) * PB + m;
#pragma unroll 8
        for (int j = 0; j < 32; j += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[j * PB], pb[j * PB], acc, 0, 0, 0);
    }
    // the four waves' partial tiles, summed in wave order; C/D layout: lane -> column lane & 31, register r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave + 4 * i;
        const float v = ((part[0][r][lane] + part[1][r][lane]) + part[2][r][lane]) + part[3][r][lane];
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = col0 + (lane & 31);
        if (dir) emit(F.gw_dir[model] + (size_t)row * kParamIn[9] + col, __builtin_fmaf(sv[row], bf[col], v));
        else emit(F.gw_final[model] + (size_t)row * 256 + col, v);
    }
}

}  // namespace nerfhip

// ================================================================================================
// C ABI
// ================================================================================================
static inline bool valid_dtype(int dtype) { return dtype == NERFHIP_F32 || dtype == NERFHIP_BF16 || dtype == NERFHIP_BF16_F8; }
static inline int compute_prec(int dtype) { return dtype == NERFHIP_F32 ? NERFHIP_F32 : NERFHIP_BF16; }

static int64_t act_tiles(int64_t n_points, int dtype) {
    const int64_t ppw = 32 * (dtype == NERFHIP_F32 ? 4 : 8);
    return (n_points + ppw - 1) / ppw * (ppw / 32);
}

extern "C" size_t nerfhip_mlp_dy_bytes(int64_t n_points, int dtype) {
    if (n_points < 0 || !valid_dtype(dtype)) return 0;
    if (dtype == NERFHIP_BF16_F8) return (size_t)act_tiles(n_points, dtype) * nerfhip::mlp::f8_dy_tile_bytes();
    return (size_t)act_tiles(n_points, dtype) * nerfhip::mlp::kDySlabs * 64 * (dtype == NERFHIP_BF16 ? 16 : 32);
}

// The split plan: which workgroup = which (job, point range).
// Rounds 1-3 gave every workgroup the SAME number of ring iterations: until the inner loops were software-pipelined (round 3) an
// iteration cost ten exposed LDS round trips whatever its bytes, and splits in proportion to the jobs' bytes measured 268-306 us
// against 225-232 us (e4m3, 1024 x 192 points).  The e4m3 launch still plans that way; the bf16 launch — see below — no longer.
// e4m3 kernel: ONE round of the 256 CUs (measured at 1024 x 192: 207-215 us vs 233-247 us for 384-768 workgroups, and half the
// split-K partials for the reduce kernel: 23 -> 12.5 us); bf16 one round too (every workgroup less is a 330 KB partial slab neither
// written nor re-read), fp32 two rounds (2,882 vs 3,244 us).
// Several models in one launch (a training step's fine + coarse network): the workgroups are shared out ACROSS the models — a model
// with a third of the points gets a third of the splits per job — instead of a second, short launch that cannot hide its pipeline
// fill (coarse pass alone: 0.49 of the HBM peak vs 0.62 for the fine pass).
#ifndef NERFHIP_DWF8_WGS
#define NERFHIP_DWF8_WGS 256
#endif
// Round 4: with the inner loops pipelined (round 3) an iteration's time DOES follow its bytes — per-workgroup wall clocks of the
// merged bf16 launch (tools/dw_probe.py, profiles/r04_dw_probe_call14_block_issue.txt): 0.69 / 0.91 / 0.82 / 1.05 / 1.40 / 1.63 us per iteration for
// stages of 10 / 18 / 20 / 26 / 32 / 36 KiB, i.e. ~0.3 us + 35 ns per KiB.  With equal iteration counts the skip-layer workgroups
// ran 626 us, the 256 x 256 layers 537 us and the rgb / first / sigma / dir jobs 280-430 us: the launch waited for 32 of its 256
// workgroups while a quarter of the CUs idled for a third of it.  The plan now equalises iterations x (a + b x stage KiB).
// The cost model is deliberately the coarse linear one.  A table of the per-class costs measured under a balanced plan (0.72 / 0.95 /
// 0.94 / 1.13 / 1.58 / 1.81 us per iteration for 10 / 18 / 20 / 26 / 32 / 36 KiB stages) makes every workgroup finish within 3 % of
// the others (profiles/r04_dw_probe_bf16_table_plan.txt) and the launch SLOWER: 473-480 us against 455-458 us in the same call —
// with the linear model the first-layer and dir-layer workgroups finish ~15 % early, and the bandwidth they release goes to the
// 256 x 256 and skip-layer workgroups that end the launch, whose partial slabs then do not all land in the same microseconds.
// Round 6: with the 2 x 4 wave split, the dot2 bias sums and the register-major epilogue an iteration's fixed part shrank; the sweep of
// profiles/r06_dw_plan_cost_ab.txt (one box, three alternating rounds) has 150 + 45 / KiB at 449-452 us in the step against 466-469 us for
// round 4's 300 + 35 / KiB, 100 + 50 the same, 50 + 55 and 0 + 60 (bytes-proportional) slower again.
#ifndef NERFHIP_DW_COST_A
#define NERFHIP_DW_COST_A 150
#endif
#ifndef NERFHIP_DW_COST_B
#define NERFHIP_DW_COST_B 45
#endif
#ifndef NERFHIP_DW_FOLD_SIGMA
#define NERFHIP_DW_FOLD_SIGMA 1      // the dir layer's workgroups also form the sigma head's gradient (same X section: h8 read once)
#endif
#ifndef NERFHIP_DW_MIN_ITERS
#define NERFHIP_DW_MIN_ITERS 48      // a workgroup should run at least this many ring iterations: the DEPTH-stage DMA pipeline
#endif                               // takes ~4 to fill, and every split costs a 330 KB partial slab the reduce kernel re-reads
static int dw_target_wgs(int dtype) {
    static const int env = [] {
        const char* e = getenv("NERFHIP_DW_WGS");            // experiments only
        return e ? atoi(e) : 0;
    }();
    return env > 0 ? env : (dtype == NERFHIP_BF16_F8 ? NERFHIP_DWF8_WGS : dtype == NERFHIP_BF16 ? NERFHIP_DWBF16_WGS : NERFHIP_DW_WGS);
}
// n_points[m] points of model m (m < n_models).  Fills jt (nsplit, soff, job, ntiles, njobs; the tensor pointers are the
// caller's) when non-null; returns the number of workgroups = partial slabs.
static int dw_plan(const int64_t* n_points, int n_models, int dtype, nerfhip::DwJobTable* jt, const bool* regen = nullptr) {
    using namespace nerfhip;
    using namespace nerfhip::mlp;
    const int njobs = n_models * kNumDwJobs;
    int64_t units[kDwMaxJobs];     // ring iterations of a job if it were one workgroup (f8: tile PAIRS)
    int64_t cap[kDwMaxJobs];
    int ns[kDwMaxJobs];
    int64_t cost[kDwMaxJobs];      // time of one ring iteration of the job (ns): cost_a + cost_b x (dY + X slabs of a stage)
    int total = 0;
    static const int cost_a = [] { const char* e = getenv("NERFHIP_DW_COST_A"); return e ? atoi(e) : -1; }();   // experiments only
    static const int cost_b = [] { const char* e = getenv("NERFHIP_DW_COST_B"); return e ? atoi(e) : -1; }();
    for (int j = 0; j < njobs; ++j) {
        const int64_t tiles = act_tiles(n_points[j / kNumDwJobs], dtype);
        const DwJob& jb = kDwJobs[j % kNumDwJobs];
        // (the e4m3 launch keeps equal iteration counts: with the byte-weighted plan it measured 336 us against 254 us; the fp32
        // launch has not been re-measured.  NERFHIP_DW_COST_A / _B = a + b x KiB instead, for experiments)
        const int ca = cost_a >= 0 ? cost_a : (dtype == NERFHIP_BF16 ? NERFHIP_DW_COST_A : 1);
        const int cb = cost_b >= 0 ? cost_b : (dtype == NERFHIP_BF16 ? NERFHIP_DW_COST_B : 0);
        const bool fold = NERFHIP_DW_FOLD_SIGMA;             // (bf16 since round 4; e4m3 and fp32 since round 5; into the dir job since round 6)
        // (regen[m]: model m's encoding sections — x1 of the first, the skip and the dir layer — are formed in the kernel, not fetched.
        // NERFHIP_DW_REGEN_PLAN=1 prices those jobs by the bytes they still fetch; by default the plan is the one of the saved
        // encodings — the same workgroups, hence the same fp32 summation order and bit-identical gradients in both forms — and the
        // three job classes simply finish early)
        static const bool regen_plan = [] { const char* e = getenv("NERFHIP_DW_REGEN_PLAN"); return e && atoi(e) != 0; }();
        const int enc_fetched = (regen_plan && regen && regen[j / kNumDwJobs] && jb.x1_enc != 0) ? jb.x1_slabs : 0;
        cost[j] = ca + (int64_t)cb * (jb.dy_slabs + jb.x1_slabs + jb.x2_slabs - enc_fetched + (fold && j % kNumDwJobs == kDwJobDir ? 2 : 0));
        if (cost[j] < 1) cost[j] = 1;
        units[j] = tiles / (dtype == NERFHIP_BF16_F8 ? 2 : 1);
        cap[j] = NERFHIP_DW_MIN_ITERS > 0 ? units[j] / NERFHIP_DW_MIN_ITERS : units[j];
        if (cap[j] > units[j]) cap[j] = units[j];
        if (cap[j] < 1) cap[j] = 1;
        // no workgroups of their own: the final layer (derived from the dir job's G by mlp_bwd_fold_kernel, mlp_layout.h kDwJobs) and,
        // folded, the sigma head (the dir layer's workgroups form dW_sigma too)
        if (j % kNumDwJobs == kDwJobFinal || (fold && j % kNumDwJobs == kDwJobSigma)) {
            cap[j] = 0;
            ns[j] = 0;
            continue;
        }
        ns[j] = 1;
        ++total;
    }
    // greedy: the next workgroup goes to the job whose workgroups currently run the LONGEST (iterations x time per iteration);
    // ties go to the jobs with the most bytes per iteration (the 256 x 256 layers, jobs 1..8 of a model), then to the lower index
    const int target = dw_target_wgs(dtype);
    while (total < target) {
        int best = -1;
        for (int j = 0; j < njobs; ++j) {
            if (ns[j] >= cap[j]) continue;
            if (best < 0) { best = j; continue; }
            const int64_t a = units[j] * cost[j] * ns[best], b = units[best] * cost[best] * ns[j];   // time per workgroup of j vs best
            const int jj = j % kNumDwJobs, bb = best % kNumDwJobs;
            const bool j_big = jj >= 1 && jj <= 7, b_big = bb >= 1 && bb <= 7;
            if (a > b || (a == b && j_big && !b_big)) best = j;
        }
        if (best < 0) break;
        ++ns[best];
        ++total;
    }
    if (jt) {
        int off = 0;
        for (int j = 0; j < kDwMaxJobs; ++j) {
            const int jj = j < njobs ? j : 0;
            jt->job[j] = kDwJobs[jj % kNumDwJobs];
            jt->nsplit[j] = j < njobs ? ns[j] : 0;
            jt->soff[j] = off;
            jt->ntiles[j] = act_tiles(n_points[jj / kNumDwJobs], dtype);
            jt->fold_of[j] = (j < njobs && ns[j] == 0 && j % kNumDwJobs == kDwJobSigma) ? j - kDwJobSigma + kDwJobDir : -1;
            if (j < njobs) off += ns[j];
        }
        jt->soff[kDwMaxJobs] = off;
        jt->njobs = njobs;
    }
    return total;
}

extern "C" int nerfhip_mlp_dw_splits(int64_t n_points, int dtype) {      // total (job, split) workgroups / partial slabs
    if (n_points <= 0 || !valid_dtype(dtype)) return 0;
    return dw_plan(&n_points, 1, dtype, nullptr);
}
// workspace = the launch's partial slabs, then one fold scratch (G, s) per model
static size_t dw_workspace_bytes(int nwg, int n_models) {
    return ((size_t)nwg * nerfhip::mlp::kDwSlabFloats + (size_t)n_models * nerfhip::mlp::kFoldScratchFloats) * sizeof(float);
}
extern "C" size_t nerfhip_mlp_dw_workspace_bytes(int64_t n_points, int dtype) {
    const int nwg = nerfhip_mlp_dw_splits(n_points, dtype);
    return nwg > 0 ? dw_workspace_bytes(nwg, 1) : 0;
}
extern "C" size_t nerfhip_mlp_dw_workspace_bytes_multi(const int64_t* n_points_host, int n_models, int dtype) {
    if (!n_points_host || n_models < 1 || n_models > nerfhip::kDwMaxModels || !valid_dtype(dtype)) return 0;
    for (int m = 0; m < n_models; ++m)
        if (n_points_host[m] <= 0) return 0;
    return dw_workspace_bytes(dw_plan(n_points_host, n_models, dtype, nullptr), n_models);
}

// The split plan itself (host logic, no GPU): splits_out[12 m + j] = workgroups of weight-gradient job j of model m, stage_kib_out
// (NULL ok) = KiB one ring iteration of that job moves.  Returns the number of workgroups.
extern "C" int nerfhip_mlp_dw_plan(const int64_t* n_points_host, int n_models, int dtype, int* splits_out, int* stage_kib_out) {
    if (!n_points_host || !splits_out || n_models < 1 || n_models > nerfhip::kDwMaxModels || !valid_dtype(dtype)) return NERFHIP_E_BADARG;
    for (int m = 0; m < n_models; ++m)
        if (n_points_host[m] <= 0) return NERFHIP_E_BADARG;
    nerfhip::DwJobTable jt;
    const int total = dw_plan(n_points_host, n_models, dtype, &jt);
    for (int j = 0; j < n_models * nerfhip::mlp::kNumDwJobs; ++j) {
        splits_out[j] = jt.nsplit[j];
        if (stage_kib_out) {
            int slabs = jt.job[j].dy_slabs + jt.job[j].x1_slabs + jt.job[j].x2_slabs;
            if (jt.fold_of[j] >= 0 || jt.nsplit[j] == 0) slabs = 0;                   // folded into another job's stage / derived
            for (int k = 0; k < n_models * nerfhip::mlp::kNumDwJobs; ++k)
                if (jt.fold_of[k] == j) slabs += jt.job[k].dy_slabs;
            stage_kib_out[j] = dtype == NERFHIP_F32 ? 2 * slabs : slabs;            // (the e4m3 kernel moves TWO tiles of slabs / 2 KiB each)
        }
    }
    return total;
}

extern "C" int nerfhip_mlp_bwd_multi(int n_models, const float* const* g_out_host, const float* const* out_host,
                                     const int64_t* n_host, const void* const* packed_bwd_host, const void* const* acts_host,
                                     void* const* dys_host, void* dw_workspace, float* const* grad_w_host,
                                     float* const* grad_b_host, int accumulate, int dtype, int phases, const float* g_scale,
                                     const nerfhip_adam_fused* adam, nerfhip_stream_t stream) {
    return nerfhip_mlp_bwd_multi_rays(n_models, g_out_host, out_host, n_host, packed_bwd_host, acts_host, dys_host, dw_workspace, grad_w_host,
                                      grad_b_host, accumulate, dtype, phases, g_scale, adam, nullptr, stream);
}

extern "C" int nerfhip_mlp_bwd_multi_rays(int n_models, const float* const* g_out_host, const float* const* out_host,
                                          const int64_t* n_host, const void* const* packed_bwd_host, const void* const* acts_host,
                                          void* const* dys_host, void* dw_workspace, float* const* grad_w_host,
                                          float* const* grad_b_host, int accumulate, int dtype, int phases, const float* g_scale,
                                          const nerfhip_adam_fused* adam, const nerfhip_enc_source* enc, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n_models >= 1 && n_models <= nerfhip::kDwMaxModels);
    if (!valid_dtype(dtype)) return NERFHIP_E_UNSUPPORTED;
    bool regen[nerfhip::kDwMaxModels] = {false, false};
    if (enc) {
        // the encodings are formed in the bf16 weight-gradient kernel's dir + sigma job class only
        if (dtype != NERFHIP_BF16 || !NERFHIP_DW_FOLD_SIGMA || !NERFHIP_DW_BLOCKED) return NERFHIP_E_UNSUPPORTED;
        for (int m = 0; m < n_models; ++m) {
            if (!enc->rays[m]) continue;                              // (this model's encodings were saved)
            NERFHIP_CHECK_ARG(enc->z[m] && enc->S[m] > 0 && enc->S[m] % 32 == 0 && n_host[m] % 256 == 0 && n_host[m] % enc->S[m] == 0);
            NERFHIP_CHECK_ARG(n_host[m] / 32 < (int64_t)1 << 31);
            if ((((uintptr_t)enc->rays[m]) | ((uintptr_t)enc->z[m])) & 15) return NERFHIP_E_ALIGN;
            regen[m] = true;
        }
    }
    NERFHIP_CHECK_ARG(g_out_host && out_host && n_host && packed_bwd_host && acts_host && dys_host && grad_w_host && grad_b_host);
    NERFHIP_CHECK_ARG(!(adam && accumulate));
    nerfhip::GradTable G;
    nerfhip::DwJobTable jt;
    for (int m = 0; m < n_models; ++m) {
        NERFHIP_CHECK_ARG(n_host[m] > 0);           // (an empty batch launches nothing: the single-model entry point handles it)
        NERFHIP_CHECK_ARG(g_out_host[m] && out_host[m] && packed_bwd_host[m] && acts_host[m] && dys_host[m] && dw_workspace);
        if ((((uintptr_t)g_out_host[m]) | ((uintptr_t)out_host[m]) | ((uintptr_t)packed_bwd_host[m]) | ((uintptr_t)acts_host[m]) |
             ((uintptr_t)dys_host[m])) & 15)
            return NERFHIP_E_ALIGN;
    }
    const int nwg = dw_plan(n_host, n_models, dtype, &jt, regen);
    for (int m = 0; m < nerfhip::kDwMaxModels; ++m) {
        const bool on = m < n_models && regen[m];
        jt.enc_rays[m] = on ? enc->rays[m] : nullptr;
        jt.enc_z[m] = on ? enc->z[m] : nullptr;
        jt.enc_tpr[m] = on ? enc->S[m] / 32 : 1;
    }
    float* const fold_scratch = (float*)dw_workspace + (size_t)nwg * nerfhip::mlp::kDwSlabFloats;
    nerfhip::FoldArgs F;
    for (int m = 0; m < nerfhip::kDwMaxModels; ++m) {
        const int mm = m < n_models ? m : 0;
        F.image[m] = reinterpret_cast<const float*>((const uint8_t*)packed_bwd_host[mm] +
                                                    (size_t)nerfhip::mlp::bwd_padded_pieces(compute_prec(dtype)) * nerfhip::mlp::kPieceBytes);
        F.gw_final[m] = grad_w_host[12 * mm + 8];
        F.gb_final[m] = grad_b_host[12 * mm + 8];
        F.gw_dir[m] = grad_w_host[12 * mm + 9];
    }
    for (int j = 0; j < nerfhip::kDwMaxJobs; ++j) {
        const int jj = j < jt.njobs ? j : 0, m = jj / nerfhip::mlp::kNumDwJobs, prm = nerfhip::mlp::kDwJobs[jj % nerfhip::mlp::kNumDwJobs].param;
        NERFHIP_CHECK_ARG(grad_w_host[12 * m + prm] && grad_b_host[12 * m + prm]);
        G.w[j] = grad_w_host[12 * m + prm];
        G.b[j] = grad_b_host[12 * m + prm];
        jt.acts[j] = (const uint8_t*)acts_host[m];
        jt.dys[j] = (const uint8_t*)dys_host[m];
    }
    nerfhip::AdamFused A;
    A.state = nullptr;
    A.lr = A.beta1 = A.beta2 = A.eps = A.wd = 0.f;
    for (int m = 0; m < nerfhip::kDwMaxModels; ++m) { A.param[m] = A.m[m] = A.v[m] = nullptr; A.grad0[m] = nullptr; }
    if (adam) {
        NERFHIP_CHECK_ARG(adam->state && adam->n_models == n_models);
        for (int m = 0; m < n_models; ++m) {
            NERFHIP_CHECK_ARG(adam->param[m] && adam->exp_avg[m] && adam->exp_avg_sq[m] && adam->grad_flat[m]);
            A.param[m] = adam->param[m]; A.m[m] = adam->exp_avg[m]; A.v[m] = adam->exp_avg_sq[m]; A.grad0[m] = adam->grad_flat[m];
        }
        A.state = adam->state; A.lr = adam->lr; A.beta1 = adam->beta1; A.beta2 = adam->beta2; A.eps = adam->eps; A.wd = adam->weight_decay;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 rgrid(8 * (nerfhip::mlp::kDwMaxXTiles + 1), (unsigned)jt.njobs);
    const bool do_chain = phases & 1, do_dw = phases & 2, do_reduce = phases & 4;
    if (do_chain) {                                 // ONE launch for the chains of all models (fine first: the long one leads)
        int64_t tiles[nerfhip::kDwMaxModels];
        for (int m = 0; m < n_models; ++m) tiles[m] = act_tiles(n_host[m], dtype);
        nerfhip::launch_bwd_chain(n_models, g_out_host, g_scale, out_host, n_host, packed_bwd_host, acts_host, dys_host, dtype, tiles, s);
    }
    if (do_dw) {
        if (dtype == NERFHIP_BF16_F8)
            hipLaunchKernelGGL(nerfhip::mlp_bwd_dw_f8_kernel, dim3(nwg), dim3(512), 0, s, jt, (float*)dw_workspace);
        else if (dtype == NERFHIP_BF16)
            hipLaunchKernelGGL(nerfhip::mlp_bwd_dw_kernel<NERFHIP_BF16>, dim3(nwg), dim3(512), 0, s, jt, (float*)dw_workspace);
        else
            hipLaunchKernelGGL(nerfhip::mlp_bwd_dw_kernel<NERFHIP_F32>, dim3(nwg), dim3(512), 0, s, jt, (float*)dw_workspace);
    }
    if (do_reduce) {
        if (dtype == NERFHIP_BF16_F8)
            hipLaunchKernelGGL(nerfhip::mlp_bwd_reduce_kernel<true>, rgrid, dim3(256), 0, s, jt, (const float*)dw_workspace, G, fold_scratch,
                               accumulate, A);
        else
            hipLaunchKernelGGL(nerfhip::mlp_bwd_reduce_kernel<false>, rgrid, dim3(256), 0, s, jt, (const float*)dw_workspace, G, fold_scratch,
                               accumulate, A);
        hipLaunchKernelGGL(nerfhip::mlp_bwd_fold_kernel, dim3(nerfhip::kFoldBlocks, (unsigned)n_models), dim3(256), 0, s, F,
                           (const float*)fold_scratch, accumulate, A);
    }
    return nerfhip_launch_status();
}

#if NERFHIP_DW_PROBE
// debug builds only (not part of include/nerfhip.h): the dW kernels' per-wave cycle accounts of the last launch
extern "C" int nerfhip_debug_dw_probe(unsigned* host_dst, int n_words) {
    if (!host_dst || n_words < 0 || n_words > 1024 * 8 * 8) return NERFHIP_E_BADARG;
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(nerfhip::g_dw_probe), (size_t)n_words * 4, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -100;
}
#endif

extern "C" int nerfhip_mlp_bwd_phases(const float* g_out, const float* out, int64_t n, const void* packed_bwd, const void* acts,
                                      void* dys, void* dw_workspace, float* const* grad_w_host, float* const* grad_b_host,
                                      int accumulate, int dtype, int phases, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0);
    if (!valid_dtype(dtype)) return NERFHIP_E_UNSUPPORTED;
    NERFHIP_CHECK_ARG(grad_w_host && grad_b_host);
    for (int i = 0; i < 12; ++i) NERFHIP_CHECK_ARG(grad_w_host[i] && grad_b_host[i]);
    if (n == 0) return 0;
    return nerfhip_mlp_bwd_multi(1, &g_out, &out, &n, &packed_bwd, &acts, &dys, dw_workspace, grad_w_host, grad_b_host, accumulate,
                                 dtype, phases, nullptr, nullptr, stream);
}

extern "C" int nerfhip_mlp_bwd(const float* g_out, const float* out, int64_t n, const void* packed_bwd, const void* acts,
                               void* dys, void* dw_workspace, float* const* grad_w_host, float* const* grad_b_host,
                               int accumulate, int dtype, nerfhip_stream_t stream) {
    return nerfhip_mlp_bwd_phases(g_out, out, n, packed_bwd, acts, dys, dw_workspace, grad_w_host, grad_b_host, accumulate, dtype, 7,
                                  stream);
}
