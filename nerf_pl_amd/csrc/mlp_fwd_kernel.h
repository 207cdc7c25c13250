// K2: fused NeRF MLP forward — replaces, per `inference` call, the point-chunk loop of
// reference models/rendering.py:115-141 (embedding_xyz -> cat -> model) together with
// Embedding.forward (models/nerf.py:21-38) and NeRF.forward (models/nerf.py:83-124):
// ~70 ATen launches and ~5 KB/point of HBM round trips per chunk become one launch whose only HBM
// traffic is 4 B in (z) + 16 B out (rgb,sigma) per point plus the L2-resident weight stream.
//
// Mapping (MFMA-bound): one wavefront owns 32 points for the whole network; activations
// never leave its registers (C/D fragment of layer l == B operand of layer l+1, see mlp_layout.h);
// the weights — pre-packed in A-fragment order — are DMA'd global->LDS (global_load_lds_dwordx4,
// lane-linear 1 KiB pieces) into a 3-slot x 32 KiB ring shared by the workgroup's waves, with one
// s_barrier per 32 KiB chunk and two chunks always in flight (counted vmcnt, never drained to 0).
// Layers run output tile by output tile, software-pipelined: the epilogue of tile t is emitted between the
// MFMAs of tile t+1 (see "Software-pipelined layer" below).
//   bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate, 8 waves (2 per SIMD) = 256 points / workgroup
//   fp32: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), 4 waves (1 per SIMD) = 128 points / workgroup
#pragma once
#include <type_traits>

#include "common.h"
#include "mlp_layout.h"
#include "f8_store.h"
#include "sampling_math.h"

#ifndef NERFHIP_STORE_AUX
#define NERFHIP_STORE_AUX 2     // cache-policy bits of the activation stores: 2 = nt (written once, read by another kernel: -7 %)
#endif
#ifndef NERFHIP_FAST_SINCOS
#define NERFHIP_FAST_SINCOS 1   // bf16 kernels only; the fp32 (parity) kernels always use sincosf
#endif
#ifndef NERFHIP_EXP
#define NERFHIP_EXP 0
#endif
#ifndef NERFHIP_DMA_SADDR
#define NERFHIP_DMA_SADDR 1     // weight-stream DMAs address as SGPR base + one constant per-lane VGPR offset (no per-piece VALU address)
#endif
#ifndef NERFHIP_PF2
#define NERFHIP_PF2 2      // prefetch depth of the 2-waves-per-SIMD (256-register) bf16 kernels
#endif

namespace nerfhip {
using namespace mlp;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PREC> struct PrecTraits;
template <> struct PrecTraits<NERFHIP_BF16> {
    using Slab = bf16x8;                 // 8 input features of one point (4 VGPRs)
};
template <> struct PrecTraits<NERFHIP_F32> {
    using Slab = f32x8;                  // 8 VGPRs
};

// Launch geometry.  bf16: 8 waves (2 per SIMD, 256 regs each) = 256 points / workgroup: the second wave of a SIMD
// fills the matrix pipe while the first waits (LDS, chunk barrier) or issues its epilogue VALU.  fp32: 4 waves (1 per
// SIMD, 512 regs; an fp32 slab set is 128 registers).  (Round 1 measured a 4-wave/512-register bf16 activation-saving
// build as bimodal across MI355X boxes — 376 us on some, ~900 us on others, same binary — hence 8 waves everywhere.)
template <int PREC, bool SAVE> struct KCfg {
    static constexpr int NW = (PREC == NERFHIP_BF16) ? 8 : 4;
    static constexpr int WPS = (PREC == NERFHIP_BF16) ? 2 : 1;
};

__device__ __forceinline__ void make_slab(bf16x8& s, const float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = (__bf16)v[j];
}
__device__ __forceinline__ void make_slab(f32x8& s, const float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = v[j];
}

// one 16-byte-per-lane global->LDS DMA; LDS destination = wave-uniform `lds_dst` + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// the same with the source as wave-uniform base (SGPR pair) + per-lane byte offset `voff`
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

template <int PREC, int NCH, bool COUNT_STORES = false>
struct WeightStream {
    static constexpr int NW = KCfg<PREC, COUNT_STORES>::NW;    // COUNT_STORES == SAVE variant
    static constexpr int LPW = kChunkPieces / NW;   // DMA instructions per wave per chunk
    const uint8_t* gsrc;     // packed + lane*16 | NERFHIP_DMA_SADDR: packed (wave-uniform)
    unsigned voff;           // NERFHIP_DMA_SADDR: lane*16
    unsigned lds_base;       // LDS byte address of the ring
    int wave;                // wave index in the workgroup (SGPR)
    int pending;             // vector-memory STORE instructions issued since the last boundary (SAVE variant).
                             // Straight-line code: the optimiser folds this to a constant at every boundary.
    int pending_prev;        // ... and in the interval before that
#if NERFHIP_STREAM_PROBE
    unsigned pr_wait = 0, pr_bar = 0, pr_n = 0;     // 10 ns ticks at the boundaries' s_waitcnt / s_barrier, boundaries passed
#endif

    __device__ __forceinline__ void issue_piece(int c, int k) const {      // k-th of this wave's LPW pieces of chunk c
        const int piece = wave + k * NW;
#if NERFHIP_DMA_SADDR
        glds16_s(gsrc + ((size_t)c * kChunkPieces + piece) * kPieceBytes, voff,
                 lds_base + (unsigned)((c % kSlots) * kChunkBytes + piece * kPieceBytes));
#else
        glds16(gsrc + ((size_t)c * kChunkPieces + piece) * kPieceBytes,
               lds_base + (unsigned)((c % kSlots) * kChunkBytes + piece * kPieceBytes));
#endif
    }
    __device__ __forceinline__ void issue_chunk(int c) const {
#pragma unroll
        for (int i = 0; i < LPW; ++i) issue_piece(c, i);
    }
    // Called once for EVERY piece index G of the stream, in increasing order, right before piece G is read: the
    // first piece of a chunk is the chunk boundary.  (Measured: spreading the LPW refill DMAs over the chunk and
    // staggering them between the two halves of the workgroup — instead of one burst behind the barrier — is
    // SLOWER: 203 vs 184 us forward, and 3x on the 4-wave SAVE variant; the burst stays.)
    template <int G>
    __device__ __forceinline__ void at_piece() {
        if constexpr (G % kChunkPieces == 0) boundary(G / kChunkPieces);
    }
    // Called by every wave right before the first piece of chunk c is read.
    __device__ __forceinline__ void boundary(int c) {
#if NERFHIP_EXP == 1          // timing experiment only (results invalid): barriers kept, no refill DMAs after the prologue
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return;
#elif NERFHIP_EXP == 2        // timing experiment only: neither barriers nor refills (pure MFMA + LDS reads + epilogues)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        return;
#endif
        // (1) my DMAs for chunk c have landed (chunk c+1's may stay in flight), my LDS reads of chunk c-1
        // have returned; (2) barrier: same holds for every wave => chunk c is readable and the slot of
        // chunk c-1 is free; (3) refill that slot with chunk c+2.
        // vmcnt retires in issue order and counts stores too: the ops younger than chunk c's DMAs are the
        // LPW DMAs of chunk c+1 plus the `pending` activation stores issued since the previous boundary
        // (older stores are waited for as well — harmless).  Under-counting only over-waits.
        if constexpr (COUNT_STORES) {
            // chunk c's DMAs were issued at boundary c-2; younger than them are the stores of the interval before the previous
            // boundary (pending_prev), chunk c+1's DMAs and the stores since the previous boundary (pending)
            const int n = (c + 1 < NCH ? LPW : 0) + pending + (NERFHIP_STORE_SLACK ? pending_prev : 0);
            pending_prev = pending;
            pending = 0;
#if NERFHIP_STREAM_PROBE
            const unsigned t0 = (unsigned)__builtin_amdgcn_s_memrealtime();
            wait_only(n);
            const unsigned t1 = (unsigned)__builtin_amdgcn_s_memrealtime();
            asm volatile("s_barrier" ::: "memory");
            const unsigned t2 = (unsigned)__builtin_amdgcn_s_memrealtime();
            pr_wait += t1 - t0; pr_bar += t2 - t1; pr_n += 1;
#else
            wait_barrier(n);
#endif
        } else if (c + 1 < NCH) {
            if (LPW == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else          asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (c + 2 < NCH) issue_chunk(c + 2);
    }
#if NERFHIP_STREAM_PROBE
    static __device__ __forceinline__ void wait_only(int n) {
#define NH_WB(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); break;
        switch (n < 0 ? 0 : (n > 48 ? 48 : (n <= 8 ? n : (n & ~3)))) {
            NH_WB(0) NH_WB(1) NH_WB(2) NH_WB(3) NH_WB(4) NH_WB(5) NH_WB(6) NH_WB(7) NH_WB(8)
            NH_WB(12) NH_WB(16) NH_WB(20) NH_WB(24) NH_WB(28) NH_WB(32) NH_WB(36) NH_WB(40) NH_WB(44) NH_WB(48)
            default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        }
#undef NH_WB
    }
#endif
    static __device__ __forceinline__ void wait_barrier(int n) {
#define NH_WB(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        switch (n < 0 ? 0 : (n > 48 ? 48 : (n <= 8 ? n : (n & ~3)))) {   // multiples of 4 above 8 (round DOWN = safe)
            NH_WB(0) NH_WB(1) NH_WB(2) NH_WB(3) NH_WB(4) NH_WB(5) NH_WB(6) NH_WB(7) NH_WB(8)
            NH_WB(12) NH_WB(16) NH_WB(20) NH_WB(24) NH_WB(28) NH_WB(32) NH_WB(36) NH_WB(40) NH_WB(44) NH_WB(48)
            default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        }
#undef NH_WB
    }
};

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) — guarantees static register indexing
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ---- training: save B-operand slabs in register (fragment) order, one coalesced 16-B store/lane/piece ----
// Buffer stores through a per-wave descriptor with a 32-bit per-lane offset VGPR (no 64-bit address VGPR
// pairs competing with the accumulators).  The section offset is added to the VOFFSET and soffset stays the
// constant 0: with a wave-uniform soffset in an SGPR, LLVM's hazard recognizer assumes the ">64-bit store data
// followed by a VALU write of the data VGPR" hazard cannot occur and lets the next VALU instruction overwrite
// v[d:d+3] right behind the store — on gfx950 that corrupts lanes 12-15 of every 16 (measured: dY slabs with
// 0x4000 patterns from the following v_and).  With soffset = 0 the compiler inserts the wait states.
template <int PREC, int NCH, typename Slab, bool CS>
__device__ __forceinline__ void save_slabs(WeightStream<PREC, NCH, CS>& st, uint8_t* tile_ptr, int sec,
                                           const Slab* slabs, int n, int lane) {
    constexpr int IL = act_il(PREC);        // (non-F8 callers only) piece pitch of the interleaved block, mlp_layout.h
    // One descriptor per call with the section offset folded into its (wave-uniform, SALU-computed) base: every store
    // then uses the SAME voffset VGPR (lane * sizeof(Slab)) and a small immediate.  (Folding the section offset into
    // the voffset instead made hipcc hoist ~40 distinct per-lane offset VGPRs to the top of the kernel: +80 live
    // registers, spills.)
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_ptr + (size_t)sec * 64 * sizeof(Slab) * IL, 0,
                                                                  (int)(n * 64 * sizeof(Slab) * IL), 0x00020000);
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(Slab);
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const u32x4* src = reinterpret_cast<const u32x4*>(&slabs[i]);
#pragma unroll
        for (int q = 0; q < (int)(sizeof(Slab) / 16); ++q) {
            __builtin_amdgcn_raw_buffer_store_b128(src[q], rs, voff + (unsigned)(i * 64 * sizeof(Slab) * IL) + 16 * q, 0, NERFHIP_STORE_AUX);
            st.pending += 1;
        }
    }
}

// The lane id, recomputed (v_mbcnt): a value the register allocator can drop and recreate instead of keeping the prologue's
// copy — and everything derived from it — live (at the 256-register limit: spilled to scratch) across the whole network.
__device__ __forceinline__ int fresh_lane() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// the same as opaque inline asm (the builtins are CSE'd back into one long-lived value): for the single use after the last
// layer; inside the layers an asm statement would act as a scheduling barrier (measured: +30 spilled registers)
__device__ __forceinline__ int fresh_lane_opaque() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// ====================================================================================================================
// Software-pipelined layer.  A layer runs output tile by output tile: acc = bias; acc += W_frag(t, ks) * B[ks] over the
// input slabs ks (weights packed in the same (t, ks) order, mlp_pack.hip).  The epilogue of tile t (activation, pack into the
// next layer's B slabs 2t, 2t+1, gate bits, stores) is EMITTED, in eight 2-value pieces, between the first MFMAs of tile
// t+1 — which accumulates into the other accumulator — and the bias of tile t+2 is read into the freed accumulator a few
// MFMAs later; a layer's last tile is finished inside the next layer's first tile (its slabs 2(NT-1), 2(NT-1)+1 are not
// consumed before slab step enc_slabs + 2(NT-1)).  A `sched_barrier` after every MFMA step pins that order.
// (Left to itself hipcc keeps a tile's epilogue — s_nop + ~50 VALU + 4 bias reads + wait — in one block BETWEEN the tiles
// and merges the two accumulators into one register set: every 16 MFMAs each wave, and — the waves of a SIMD being
// chunk-synchronised by the ring barriers — the whole SIMD, left the matrix pipe idle for ~400 cycles.  Same-call A/B
// at 1024x192 points: inference 182-188 -> 171-173 us, activation-saving e4m3 variant 281 -> 260-267 us.)
// Biases are read from a workgroup-shared 12 KiB LDS image filled once in the prologue (the in-stream bias pieces still
// travel through the ring; they are not read), A fragments and encoding operands are prefetched D steps ahead ACROSS
// layer boundaries (fragment slots are numbered over the whole network).
typedef __attribute__((ext_vector_type(2))) short nh_s16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 nh_bf16x2;

NH_HD constexpr int layer_frags(int L) { return L < 0 ? 0 : kLayers[L].nt * (kLayers[L].enc_slabs + kLayers[L].chain_slabs); }
NH_HD constexpr int frag_base(int L) {            // fragments before layer L (execution order)
    int n = 0;
    for (int i = 0; i < L; ++i) n += layer_frags(i);
    return n;
}
// pending-epilogue kinds
constexpr int PK_NONE = 0, PK_RELU = 1, PK_LINEAR = 2, PK_SIGMA = 3;
// per-layer facts (execution-order index, kLayers)
NH_HD constexpr bool layer_relu(int L) { return L <= 7 || L == kDirLayer; }
NH_HD constexpr int pend_kind(int PL) {
    return PL < 0 ? PK_NONE : (PL == kSigmaLayer ? PK_SIGMA : (layer_relu(PL) ? PK_RELU : PK_LINEAR));
}
// saved-activation section / gate piece of a layer's OUTPUT; section a layer stores in the fp8 mode (its chain INPUT)
NH_HD constexpr int layer_out_sec(int L) { return L <= 7 ? act_h(L + 1) : (L == kDirLayer ? kActT : -1); }
NH_HD constexpr int layer_gate_piece(int L) { return L <= 7 ? mask_piece_h(L + 1) : (L == kDirLayer ? kMaskPieceT : -1); }
NH_HD constexpr int layer_in_sec(int L) {       // (the dir layer runs on h8 through the folded final layer: mlp_layout.h kLayers)
    return (L >= 1 && L <= 7) ? act_h(L) : (L == kDirLayer ? act_h(8) : (L == kDirLayer + 1 ? kActT : -1));
}
// sections the activation-saving forward writes
NH_HD constexpr bool layer_out_saved(int L) { return layer_out_sec(L) >= 0; }

#ifndef NERFHIP_EXP_NOPS
#define NERFHIP_EXP_NOPS 0
#endif
#ifndef NERFHIP_EXP_SMALL
#define NERFHIP_EXP_SMALL 0     // code-size experiment only (results invalid): no gate words, no running maximum
#endif
#ifndef NERFHIP_SAVE_BURST
#define NERFHIP_SAVE_BURST 1      // 1, 2, 4, 8: divides the 8 / 16 output slabs of every saved layer
#endif
#ifndef NERFHIP_PF_SAVE
#define NERFHIP_PF_SAVE 2
#endif
template <int PREC, int SV, typename Slab>
struct PipeCtx {
    static constexpr int D = (PREC != NERFHIP_BF16) ? 1 : (SV != 0 ? NERFHIP_PF_SAVE : NERFHIP_PF2);
    Slab a[D];        // A-fragment ring
    Slab bq[D];       // encoding-operand ring (slots of fragments whose B operand is an encoding slab)
    f32x16 acc[2];
    const char* smem_lane;     // ring + lane*16
    const char* bias_lane;     // bias image + (lane>>5)*16
    const char* enc_x;         // this lane's slot of the wave's encoding stash
    const char* enc_d;
    // activation-saving variants
    uint8_t* tile;             // this wave's saved-activation block: slab / pair sections are addressed from here,
    uint8_t* gate_base;        // its gate pieces from here and
    uint8_t* scale_base;       // its scale dwords from here (three bases: the looped layer triples shift each by its own amount)
    unsigned gw[4];            // ReLU gate words of the layer in flight
    unsigned mx;               // fp8 storage: this lane's max |output| of the layer in flight, as two bf16 halves (even / odd values)
    int sb;                    // fp8 storage: e8m0 scale byte of the most recently finished section
};

// Gate bits of one packed dword of two post-ReLU bf16 values (mlp_layout.h gate_bit: dword k of a gate word -> bit 15-k of each
// half-word): the halves are >= 0, so min(half, 1) = [half > 0]; both half-words of the gate word are shifted left by one and
// the new bits added — two packed 16-bit VALU operations.  (Inline asm: written in C, hipcc canonicalises min(x, 1) to a
// compare + select per half plus a v_perm, seven instructions and 50 bytes of code per dword.)
__device__ __forceinline__ unsigned gate_shift_in(unsigned gw, unsigned d) {
    unsigned t, r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(t) : "v"(d));
    asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(gw), "v"(t));
    return r;
}

// one 2-value piece p (0..7) of the epilogue of the finished tile `c` (tile pt of layer PL):
// -> dword (p & 3) of slab 2*pt + (p >> 2) of `po`; SAVE: gate bits, slab / gate-word stores, running maximum
template <int PREC, int SV, int PL, typename Ctx, typename St, typename Slab>
__device__ __forceinline__ void epi_piece(Ctx& cx, St& st, const f32x16& c, Slab* po, int pt, int p) {
    constexpr bool RELU = layer_relu(PL), SAVE = SV != 0, F8 = SV == 2;
    Slab& o = po[2 * pt + (p >> 2)];
    unsigned& gw = cx.gw[(pt >> 1) & 3];
    if constexpr (PREC == NERFHIP_BF16) {
        typedef __attribute__((ext_vector_type(2))) float f32x2v;
        typedef __attribute__((ext_vector_type(2))) unsigned short u16x2v;
        const f32x2v xv = {c[2 * p], c[2 * p + 1]};
        nh_bf16x2 pk = __builtin_convertvector(xv, nh_bf16x2);       // one v_cvt_pk_bf16_f32
        if (RELU) {     // relu after rounding == rounding after relu; packed signed-integer max with 0 clears negative halves
            nh_s16x2 sv = __builtin_bit_cast(nh_s16x2, pk);
            const nh_s16x2 z = {0, 0};
            sv = __builtin_elementwise_max(sv, z);
            pk = __builtin_bit_cast(nh_bf16x2, sv);
        }
        o[2 * (p & 3)] = pk[0];
        o[2 * (p & 3) + 1] = pk[1];
        if (SAVE && RELU && !NERFHIP_EXP_SMALL) gw = gate_shift_in(gw, __builtin_bit_cast(unsigned, pk));
        if (F8 && !NERFHIP_EXP_SMALL) {       // running maximum of the STORED magnitudes: non-negative bf16 halves order like unsigned integers
            unsigned d = __builtin_bit_cast(unsigned, pk);
            if (!RELU) d &= 0x7fff7fffu;
            cx.mx = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2v, cx.mx), __builtin_bit_cast(u16x2v, d)));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float x = c[2 * p + k];
            // max(bits, 0) as signed integers == relu for every non-NaN float (negative floats are negative integers)
            const float r = RELU ? __int_as_float(max(__float_as_int(x), 0)) : x;
            o[2 * (p & 3) + k] = r;
            if (SAVE && RELU) {     // bit 31 of (bits + 0x7fffffff) is [r > 0] for r >= +0
                const int idx = 8 * (2 * pt + (p >> 2)) + 2 * (p & 3) + k;
                if ((idx & 31) == 0) gw = 0u;
                gw |= ((__float_as_uint(r) + 0x7fffffffu) >> 31) << gate_bit(idx);
            }
        }
    }
    if constexpr (SAVE) {
        const int lane = fresh_lane();
        // NERFHIP_SAVE_BURST slabs per run of back-to-back stores (1 = each slab as soon as it is packed; a layer's slabs stay in
        // registers as the next layer's operands, so holding a run back costs no register)
        if (!F8 && layer_out_saved(PL) && (p & 3) == 3) {
            const int si = 2 * pt + (p >> 2);
            if ((si + 1) % NERFHIP_SAVE_BURST == 0)
                save_slabs(st, cx.tile, layer_out_sec(PL) + si + 1 - NERFHIP_SAVE_BURST, &po[si + 1 - NERFHIP_SAVE_BURST], NERFHIP_SAVE_BURST, lane);
        }
        if (RELU && p == 7 && (pt & 1)) {
            __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(cx.gate_base + layer_gate_piece(PL) * kPieceBytes * act_il(PREC, F8), 0,
                                                                           kPieceBytes, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(gw, grs, (unsigned)lane * 16u + 4u * (unsigned)(pt >> 1), 0, 0);
            st.pending += 1;
        }
    }
}

// fragment j of layer L -> ring slot; issues the LDS reads (and the chunk-boundary protocol of its pieces)
template <int PREC, int NCH, int L, int J, typename Ctx, typename St>
__device__ __forceinline__ void pipe_prefetch(Ctx& cx, St& st) {
    using Slab = typename PrecTraits<PREC>::Slab;
    constexpr Layer ly = kLayers[L];
    constexpr int NKS = ly.enc_slabs + ly.chain_slabs;
    constexpr int G0 = layer_start(L, PREC), PPF = ppf(PREC);
    constexpr int slot = (frag_base(L) + J) % Ctx::D;
    constexpr int g = G0 + J * PPF;
    auto piece_off = [](int gg) { return ((gg / kChunkPieces) % kSlots) * kChunkBytes + (gg % kChunkPieces) * kPieceBytes; };
    if constexpr (L == kLoopSecond && J == 0) {
        // the bias block's chunks travel through the ring unread: run their boundaries (refills) all the same
        static_for<0, bias_block_pieces(PREC) / kChunkPieces>([&](auto kc) {
            st.template at_piece<bias_block_start(PREC) + decltype(kc)::value * kChunkPieces>();
        });
    }
    st.template at_piece<g>();
    if constexpr (PREC == NERFHIP_BF16) {
        cx.a[slot] = *reinterpret_cast<const bf16x8*>(cx.smem_lane + piece_off(g));
    } else {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(cx.smem_lane + piece_off(g));
        st.template at_piece<g + 1>();
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(cx.smem_lane + piece_off(g + 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) { cx.a[slot][j] = a0[j]; cx.a[slot][4 + j] = a1[j]; }
    }
    constexpr int ks = frag_slab(J, ly.nt, NKS);
    if constexpr (ks < ly.enc_slabs) {
        const char* e = (ly.kind == IN_DIR_CHAIN) ? cx.enc_d : cx.enc_x;
        cx.bq[slot] = *reinterpret_cast<const Slab*>(e + ks * 64 * (int)sizeof(Slab));
    }
}
template <int L, int T, typename Ctx>
__device__ __forceinline__ void pipe_bias(Ctx& cx, f32x16& c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(cx.bias_lane + L * kPieceBytes + (32 * T + 8 * q) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) c[4 * q + k] = b[k];
    }
}

// L: this layer; NL: next layer in execution order (-1: none); PAR0: accumulator parity of tile 0; PL: the layer whose last
// tile is still pending (-1: none), its epilogue goes to `pout` (sigma head: c[0] -> *psig).  On entry acc[PAR0] holds the
// bias of tile 0 and the first D fragments are in flight; on exit the same holds for layer NL, and the epilogue of this layer's
// last tile is pending in acc[(PAR0 + NT - 1) & 1].
// Activation-saving variants (SV): the pieces also build the gate words and store slabs / gate words as they complete; in the
// fp8 mode a layer stores its INPUT section pair by pair a few steps after the pieces of each tile, under the scale byte that
// ONE wave reduction — placed right after the pending layer's last piece — produced.
template <int PREC, int NCH, int SV, int L, int NL, int PAR0, int PL, typename Ctx, typename St, typename Slab>
__device__ __forceinline__ void run_layer_pipe(Ctx& cx, St& st, const Slab* chain, Slab* out, Slab* pout, float* psig) {
    constexpr Layer ly = kLayers[L];
    constexpr int NT = ly.nt, NKS = ly.enc_slabs + ly.chain_slabs, N = NT * NKS, D = Ctx::D;
    constexpr int NN = layer_frags(NL);
    constexpr int PK = pend_kind(PL), PT = kLayers[PL >= 0 ? PL : 0].nt - 1;
    constexpr bool F8 = SV == 2;
    
    static_for<0, N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int t = i / NKS, ks = i % NKS;
        constexpr int slot = (frag_base(L) + i) % D;
        f32x16& c = cx.acc[(PAR0 + t) & 1];
        f32x16& cp = cx.acc[(PAR0 + t + 1) & 1];          // the previous tile's accumulator (pending epilogue), then tile t+1's

        Slab bs;
        if constexpr (ks < ly.enc_slabs) bs = cx.bq[slot];
        else bs = chain[ks - ly.enc_slabs];
        if constexpr (PREC == NERFHIP_BF16) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cx.a[slot], bs, c, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(cx.a[slot][j], bs[j], c, 0, 0, 0);
        }
        // refill the slot: fragment i+D of this layer, or of the next one
        if constexpr (i + D < N) pipe_prefetch<PREC, NCH, L, i + D>(cx, st);
        else if constexpr (NL >= 0 && i + D - N < NN) pipe_prefetch<PREC, NCH, (NL >= 0 ? NL : 0), i + D - N>(cx, st);

        // ---- pending epilogue of the previous tile, spread over steps ks0 .. ks0 + span - 1 ----
        constexpr int kind = (t == 0) ? PK : (layer_relu(L) ? PK_RELU : PK_LINEAR);
        constexpr int pl = (t == 0) ? (PL >= 0 ? PL : 0) : L;      // layer / tile the pending accumulator belongs to
        constexpr int pt = (t == 0) ? PT : t - 1;
        // first slab step that consumes what the pending epilogue produces (previous layer's slabs 2PT, 2PT+1)
        constexpr int ks1 = (t == 0 && (PK == PK_RELU || PK == PK_LINEAR)) ? ly.enc_slabs + 2 * PT : NKS;
        // first step: 2 where there is room (the MFMA-result hazard window has passed; at step 1 hipcc pads with s_nop)
        constexpr int ks0 = (ks1 - 2 >= 8) ? 2 : 1;
        constexpr int span = (ks1 - ks0 < 8) ? ks1 - ks0 : 8;
        static_assert(span >= 1, "no room for the pending epilogue");
        if constexpr (kind == PK_SIGMA) {
            if constexpr (ks == ks0) *psig = cp[0];
        } else if constexpr (kind != PK_NONE) {
            Slab* po = (t == 0) ? pout : out;
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (ks0 + p * span / 8 == ks) epi_piece<PREC, SV, pl>(cx, st, cp, po, pt, p);
        }
        // ---- bias of the next tile into the accumulator the epilogue just freed ----
        constexpr int KB = (ks0 + span < NKS - 1) ? ks0 + span : NKS - 1;
        if constexpr (ks == KB) {
            if constexpr (t + 1 < NT) pipe_bias<L, t + 1>(cx, cp);
            else if constexpr (NL >= 0) pipe_bias<(NL >= 0 ? NL : 0), 0>(cx, cp);
            if constexpr (F8 && t == 0 && (PK == PK_RELU || PK == PK_LINEAR)) {
                // the pending layer's output section is complete: its scale (one reduction), recorded in the tile's scale table
                const unsigned mh = cx.mx >> 16, ml = cx.mx & 0xffffu;
                cx.sb = f8_scale_byte(wave_max_u32((mh > ml ? mh : ml) << 16));
                cx.mx = 0u;
                save_scale_f8(st.pending, cx.scale_base, 0, f8_x_section(layer_out_sec(PL >= 0 ? PL : 0)), cx.sb, fresh_lane());
            }
        }
        // ---- fp8 storage: this tile's share of the layer's INPUT pairs ----
        if constexpr (F8 && PREC == NERFHIP_BF16 && layer_in_sec(L) >= 0) {
            constexpr int NP = ly.chain_slabs / 2;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (q >= t * NP / NT && q < (t + 1) * NP / NT) {
                    const int j = q - t * NP / NT;
                    const int step = (KB + 1 + j < NKS - 1) ? KB + 1 + j : NKS - 1;
                    if (step == ks) save_pair_f8(st.pending, cx.tile, layer_in_sec(L) / 2 + q, chain[2 * q], chain[2 * q + 1], cx.sb, fresh_lane());
                }
            }
        }
#if NERFHIP_EXP_NOPS      // code-size experiment only: pad every MFMA step with s_nop (tools: slow-box instruction-cache probe)
        static_for<0, NERFHIP_EXP_NOPS>([&](auto) { asm volatile("s_nop 0"); });
#endif
        __builtin_amdgcn_sched_barrier(0);
    });
}

// ---- input encodings in slot order (mlp_layout.h: enc_slot_channel) -----------------------------
// computed from the raw 3-vector: half h evaluates frequencies k = 2i+h; one sincos -> two slots
template <int F, int SLABS, typename Slab>
__device__ __forceinline__ void encode_slots(const float (&v)[3], int h, Slab* out) {
    constexpr int NPAIR = 3 * (F / 2);
    float slots[8 * SLABS];
    float vs[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) vs[c] = h ? 2.0f * v[c] : v[c];
    if constexpr (sizeof(Slab) == 16 && NERFHIP_FAST_SINCOS) {
        // bf16 kernels: hardware v_sin_f32 / v_cos_f32 (argument in REVOLUTIONS) instead of libm's sincosf, whose
        // inlined Payne-Hanek path is ~80 instructions x 42 calls per lane.  x/2pi is formed once per channel as a
        // hi+lo pair (two-constant product), scaled by the exact power of two and reduced with v_fract before the lo
        // part is added: |error| ~ 1e-6, three orders below the bf16 rounding applied to the result.
        constexpr float kInv2PiHi = 0.15915494f, kInv2PiLo = 6.4206297e-9f;
        float rh[3], rl[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rh[c] = vs[c] * kInv2PiHi;
            rl[c] = __builtin_fmaf(vs[c], kInv2PiHi, -rh[c]) + vs[c] * kInv2PiLo;
        }
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            const int i = p / 3, c = p % 3;
            const float sc = (float)(1 << (2 * i));
            const float t = __builtin_amdgcn_fractf(rh[c] * sc) + rl[c] * sc;
            slots[2 * p] = __builtin_amdgcn_sinf(t);
            slots[2 * p + 1] = __builtin_amdgcn_cosf(t);
        }
    } else {
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            const int i = p / 3, c = p % 3;
            const float arg = vs[c] * (float)(1 << (2 * i));   // x * 2^(2i+h): exact power-of-two scaling
            float s, co;
            sincosf(arg, &s, &co);
            slots[2 * p] = s;
            slots[2 * p + 1] = co;
        }
    }
#pragma unroll
    for (int idx = 2 * NPAIR; idx < 8 * SLABS; ++idx) {
        const int tail = idx - 2 * NPAIR;
        slots[idx] = (tail == 0) ? (h ? v[2] : v[0]) : (tail == 1) ? (h ? 0.0f : v[1]) : 0.0f;
    }
#pragma unroll
    for (int ks = 0; ks < SLABS; ++ks) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = slots[8 * ks + j];
        make_slab(out[ks], t8);
    }
}
// gathered from a pre-embedded row (NeRF.forward drop-in): channel of slot differs by half
template <int F, int SLABS, typename Slab>
__device__ __forceinline__ void load_slots(const float* __restrict__ row, int h, Slab* out) {
#pragma unroll
    for (int ks = 0; ks < SLABS; ++ks) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c0 = enc_slot_channel(F, SLABS, ks, 0, j), c1 = enc_slot_channel(F, SLABS, ks, 1, j);
            const int c = h ? c1 : c0;
            t8[j] = (c >= 0) ? row[c >= 0 ? c : 0] : 0.0f;
        }
        make_slab(out[ks], t8);
    }
}

constexpr int MODE_EMBEDDED = 0, MODE_RAYS = 1;

// LDS of one workgroup: bias image (one 1 KiB piece per layer, shared by the workgroup) | weight ring | per-wave input-encoding
// stash (6 slabs: the encodings are needed only by layers 0, 4 (xyz) and 10 (dir); parking them in LDS frees 24 (bf16) /
// 48 (fp32) registers)
template <int PREC, bool SAVE> struct FwdLds {
    static constexpr int kEncStash = (kXyzSlabs + kDirSlabs) * 64 * (int)sizeof(typename PrecTraits<PREC>::Slab);
    static constexpr int kBiasArea = kNumLayers * kPieceBytes;
    static constexpr int kRingOff = kBiasArea;
    static constexpr int kStashOff = kBiasArea + kSlots * kChunkBytes;
    static constexpr int kBytes = kStashOff + KCfg<PREC, SAVE>::NW * kEncStash;
};

// The whole network for the 32 * NW points of (virtual) workgroup `blk` — the body of mlp_fwd_kernel, and of every sub-pass of
// the single-launch render kernels (mlp_render_kernel.h), which call it in a loop with wave-uniform arguments.  On return every
// wave has issued its output stores (not waited for them); the weight ring / bias image may still be read by slower waves.
// SV: 0 = inference, 1 = save activations in the compute precision, 2 = save them as block-scaled e4m3 (bf16 compute)
template <int PREC, int MODE, bool SIGMA_ONLY, int SV>
__device__ __forceinline__ void mlp_fwd_body(char* const lds_all, const unsigned blk, const float* __restrict__ in0,
                                             const float* __restrict__ in1, int64_t n, int64_t aux, const uint8_t* __restrict__ packed,
                                             float* __restrict__ out, uint8_t* __restrict__ save, const FwdZGen& zg) {
    using Slab = typename PrecTraits<PREC>::Slab;
    constexpr bool SAVE = SV != 0, F8 = SV == 2;
    constexpr int NW = KCfg<PREC, SAVE>::NW;
    constexpr int NCH = SIGMA_ONLY ? chunks_upto_layer(kSigmaLayer + 1, PREC) : chunks_upto_layer(kNumLayers, PREC);
    constexpr int kEncStash = FwdLds<PREC, SAVE>::kEncStash;
    constexpr int kBiasArea = FwdLds<PREC, SAVE>::kBiasArea;
    char* const bias_area = lds_all;
    char* const ring = lds_all + kBiasArea;
    char* const stash_area = ring + kSlots * kChunkBytes;

#ifndef NERFHIP_CLOCK_PROBE
#define NERFHIP_CLOCK_PROBE 0      // debug builds: every wave of the e4m3 activation-saving variant records its shader-clock cycles and
#endif                             // 100 MHz wall ticks in spare dwords of its tile's scale piece (tools/kbench.py --clock-probe)
#if NERFHIP_CLOCK_PROBE
    const uint64_t probe_c0 = clock64(), probe_w0 = wall_clock64();
#endif
#if NERFHIP_STREAM_PROBE
    const uint64_t probe_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int h = lane >> 5;
    const int64_t p = (int64_t)blk * (32 * NW) + wave * 32 + (lane & 31);
    const bool valid = p < n;
    const int64_t pc = valid ? p : n - 1;

    // ---- raw inputs (ordinary loads, issued before the DMA stream starts) ----
    float xyz[3] = {0.f, 0.f, 0.f}, dir[3] = {0.f, 0.f, 0.f};
    const float* row = nullptr;
    if (MODE == MODE_RAYS) {
        const int64_t S = aux;
        const int64_t ray = pc / S;
        const float* rp = in0 + ray * 8;
        float zv;
        if (zg.z_out) {      // (wave-uniform) coarse depths formed here: rendering.py:183-204
            const float pr = zg.perturb > 0.0f ? zg.prand[pc] : 0.0f;
            zv = coarse_z_sample(rp[6], rp[7], (int)(pc - ray * S), (int)S, zg.use_disp, zg.perturb, pr);
            if (valid && h == 0) zg.z_out[p] = zv;
        } else {
            zv = in1[pc];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dir[c] = rp[3 + c];
            xyz[c] = nh_add(rp[c], nh_mul(dir[c], zv));   // o + d*z   rendering.py:206-207
        }
    } else {
        row = in0 + pc * aux;
    }

    WeightStream<PREC, NCH, SAVE> st;
#if NERFHIP_DMA_SADDR
    st.gsrc = packed;
#else
    st.gsrc = packed + lane * 16;
#endif
    st.voff = (unsigned)lane * 16u;
    st.lds_base = (unsigned)(uintptr_t)ring;
    st.wave = wave;
    st.pending = 0;
    st.pending_prev = 0;
    // wave-uniform base of this wave's activation block (SAVE); the store helpers derive descriptors from it
    constexpr int IL = act_il(PREC, F8);    // the saved block's pieces are IL KiB apart (mlp_layout.h: bf16 slabs 8, otherwise 1)
#ifdef NERFHIP_EXP_TILEWRAP    // timing experiment only (results invalid): every wave stores into one of a few L2-resident tile blocks
    uint8_t* tile_base = SAVE ? save + tile_block_off((long long)(((size_t)blk * NW + wave) & (NERFHIP_EXP_TILEWRAP - 1)),
                                                      F8 ? f8_act_tile_bytes() : act_tile_bytes(PREC), IL)
                              : (uint8_t*)nullptr;
#else
    uint8_t* tile_base = SAVE ? save + tile_block_off((long long)blk * NW + wave, F8 ? f8_act_tile_bytes() : act_tile_bytes(PREC), IL)
                              : (uint8_t*)nullptr;
#endif

    {
        // bias image: the bias piece of every layer this kernel runs, DMA'd once from the stream's bias block (older than
        // chunk 0's DMAs, so the first chunk boundary's vmcnt wait + barrier covers it)
        constexpr int NLY = SIGMA_ONLY ? kSigmaLayer + 1 : kNumLayers;
        static_for<0, NLY>([&](auto lc) {
            constexpr int Lb = decltype(lc)::value;
            if (wave == Lb % NW) {
#if NERFHIP_DMA_SADDR
                glds16_s(st.gsrc + (size_t)(bias_block_start(PREC) + Lb) * kPieceBytes, st.voff,
                         (unsigned)(uintptr_t)bias_area + (unsigned)(Lb * kPieceBytes));
#else
                glds16(st.gsrc + (size_t)(bias_block_start(PREC) + Lb) * kPieceBytes,
                       (unsigned)(uintptr_t)bias_area + (unsigned)(Lb * kPieceBytes));
#endif
            }
        });
    }
    st.issue_chunk(0);
    if (NCH > 1) st.issue_chunk(1);

    const char* smem_lane = ring + lane * 16;

    Slab encx[kXyzSlabs];
    Slab encd[kDirSlabs];
    if (MODE == MODE_RAYS) {
        encode_slots<10, kXyzSlabs>(xyz, h, encx);
        if (!SIGMA_ONLY) encode_slots<4, kDirSlabs>(dir, h, encd);
    } else {
        load_slots<10, kXyzSlabs>(row, h, encx);
        if (!SIGMA_ONLY) load_slots<4, kDirSlabs>(row + kXyzCh, h, encd);
    }

    if constexpr (F8) {
        if constexpr (PREC == NERFHIP_BF16) {
            // encodings: |sin|,|cos| <= 1 and scene coordinates << 448 => e4m3 under the fixed scale 2^0
            save_pair_f8(st.pending, tile_base, kActEncX / 2, encx[0], encx[1], 127, lane);
            save_pair_f8(st.pending, tile_base, kActEncX / 2 + 1, encx[2], encx[3], 127, lane);
            save_pair_f8(st.pending, tile_base, kActEncD / 2, encd[0], encd[1], 127, lane);
            save_scale_f8(st.pending, tile_base, f8_act_scale_off(), f8_x_section(kActEncX), 127, lane);
            save_scale_f8(st.pending, tile_base, f8_act_scale_off(), f8_x_section(kActEncD), 127, lane);
        }
    } else if (SAVE) {
        // (wave-uniform) the bf16 step leaves the encoding slabs out: mlp_bwd_dw_kernel forms them again from (rays, z)
        if (!(PREC == NERFHIP_BF16 && MODE == MODE_RAYS && zg.skip_enc_save)) {
            save_slabs(st, tile_base, kActEncX, encx, kXyzSlabs, lane);
            save_slabs(st, tile_base, kActEncD, encd, kDirSlabs, lane);
        }
    }
    char* enc_x = stash_area + wave * kEncStash;      // wave-uniform stash bases
    char* enc_d = enc_x + kXyzSlabs * 64 * (int)sizeof(Slab);
#pragma unroll
    for (int k = 0; k < kXyzSlabs; ++k) *reinterpret_cast<Slab*>(enc_x + lane * (int)sizeof(Slab) + k * 64 * (int)sizeof(Slab)) = encx[k];
    if (!SIGMA_ONLY) {
#pragma unroll
        for (int k = 0; k < kDirSlabs; ++k) *reinterpret_cast<Slab*>(enc_d + lane * (int)sizeof(Slab) + k * 64 * (int)sizeof(Slab)) = encd[k];
    }
    // activations ping-pong between two register slab sets (a layer reads one while its tiles fill the other)
    Slab ha[16], hb[16];
    {
        using Ctx = PipeCtx<PREC, SV, Slab>;
        Ctx cx;
        cx.smem_lane = smem_lane;
        cx.enc_x = enc_x + lane * (int)sizeof(Slab);
        cx.enc_d = enc_d + lane * (int)sizeof(Slab);
        uint8_t* const gate0 = SAVE ? tile_base + (F8 ? f8_act_gate_off() : act_mask_off(PREC) * IL) : (uint8_t*)nullptr;
        uint8_t* const scale0 = F8 ? tile_base + f8_act_scale_off() : (uint8_t*)nullptr;
        const char* const bias0 = bias_area + h * 16;
        cx.tile = tile_base;
        cx.gate_base = gate0;
        cx.scale_base = scale0;
        cx.bias_lane = bias0;
        cx.gw[0] = cx.gw[1] = cx.gw[2] = cx.gw[3] = 0u;
        cx.mx = 0u;
        cx.sb = 127;
        static_for<0, Ctx::D>([&](auto jc) { pipe_prefetch<PREC, NCH, 0, decltype(jc)::value>(cx, st); });
        pipe_bias<0, 0>(cx, cx.acc[0]);
        float sigma_p = 0.0f;
        Slab* const nul = nullptr;
        //                            L  NL  PAR0  pending layer
        run_layer_pipe<PREC, NCH, SV, 0, 1, 0, -1>(cx, st, (const Slab*)nullptr, ha, nul, &sigma_p);
        // ---- layers 1-3 and 5-7: ONE copy of the code, run twice (mlp_layout.h: same chunk phase, same ring slots; the second
        // pass differs by wave-uniform base offsets only).  Fully unrolled, the network is 56 KiB (inference) to 90 KiB
        // (activation-saving) of code against a 64 KiB instruction cache; some MI355X boxes run the 90 KiB kernel 1.5x slower
        // than others for the instruction fetches alone (profiles/README.md "Box-to-box spread").
        static_assert(kLoopFirst == 1 && kLoopSecond == 5 && kLoopLayers == 3, "loop below");
        static_assert(layer_start(kLoopFirst, PREC) % kChunkPieces == 0 && loop_chunk_shift(PREC) % kSlots == 0 &&
                          (layer_start(kLoopSecond, PREC) - layer_start(kLoopFirst, PREC)) % kChunkPieces == 0,
                      "the two layer triples must see the same chunk phase and ring slots");
        static_assert((frag_base(kLoopSecond) - frag_base(kLoopFirst)) % Ctx::D == 0, "... and the same fragment-ring slots");
        const uint8_t* const gsrc0 = st.gsrc;
        int n_pass;
        asm volatile("s_mov_b32 %0, 2" : "=s"(n_pass));          // opaque trip count: the loop must stay a loop
#pragma clang loop unroll(disable)
        for (int pass = 0; pass < n_pass; ++pass) {
            // (the store counters restart from 0 in every pass: under-counting only over-waits at the first boundaries)
            st.pending = 0;
            st.pending_prev = 0;
            run_layer_pipe<PREC, NCH, SV, 1, 2, 0, 0>(cx, st, ha, hb, ha, &sigma_p);
            run_layer_pipe<PREC, NCH, SV, 2, 3, 0, 1>(cx, st, hb, ha, hb, &sigma_p);
            run_layer_pipe<PREC, NCH, SV, 3, 4, 0, 2>(cx, st, ha, hb, ha, &sigma_p);
            if (pass == 0) {
                run_layer_pipe<PREC, NCH, SV, 4, 5, 0, 3>(cx, st, hb, ha, hb, &sigma_p);        // skip layer: [xyz | h4] -> ha
                // second pass = layers 5-7: everything the code addresses by layer index moves four layers on
                constexpr int kDL = kLoopSecond - kLoopFirst;
                st.gsrc = gsrc0 + (size_t)loop_chunk_shift(PREC) * kChunkBytes;
                cx.bias_lane = bias0 + kDL * kPieceBytes;
                if constexpr (SAVE) {
                    cx.tile = tile_base + (F8 ? 8 * kDL * kPieceBytes : 16 * kDL * 64 * (int)sizeof(Slab) * IL);
                    cx.gate_base = gate0 + kDL * kPieceBytes * IL;
                }
                if constexpr (F8) cx.scale_base = scale0 + kDL * 4;
            }
        }
        st.pending = 0;
        st.pending_prev = 0;
        st.gsrc = gsrc0;
        cx.bias_lane = bias0;
        cx.tile = tile_base;
        cx.gate_base = gate0;
        cx.scale_base = scale0;
        // (h8 is in hb)
        // sigma head (tile 64, accumulator 0); finishes h8's last tile first
        run_layer_pipe<PREC, NCH, SV, 8, (SIGMA_ONLY ? -1 : 9), 0, 7>(cx, st, hb, nul, hb, &sigma_p);
        if constexpr (SIGMA_ONLY) {
            if (valid && h == 0) out[p] = cx.acc[0][0];          // (n,1)   nerf.py:112-114
            return;
        } else {
            static_assert(kDirLayer == 9 && kNumLayers == 11, "layer sequence below");
            run_layer_pipe<PREC, NCH, SV, 9, 10, 1, 8>(cx, st, hb, ha, nul, &sigma_p);        // dir_encoding on [enc_d | h8] (final folded in) -> ha[0..7]
            run_layer_pipe<PREC, NCH, SV, 10, -1, 1, 9>(cx, st, ha, nul, ha, &sigma_p);       // rgb head, accumulator 1
            // (SAVE: output index from a recomputed lane id, otherwise the prologue's 64-bit address stays live across the network)
            const int lane_o = SAVE ? fresh_lane_opaque() : lane;
            const int64_t po = (int64_t)blk * (32 * NW) + wave * 32 + (lane_o & 31);
            if (po < n && (lane_o >> 5) == 0) {
                float4 o;
                o.x = 1.0f / (1.0f + expf(-cx.acc[1][0]));            // sigmoid   nerf.py:79-81
                o.y = 1.0f / (1.0f + expf(-cx.acc[1][1]));
                o.z = 1.0f / (1.0f + expf(-cx.acc[1][2]));
                o.w = sigma_p;                                        // cat([rgb, sigma])   nerf.py:122
                reinterpret_cast<float4*>(out)[po] = o;
            }
#if NERFHIP_STREAM_PROBE
            if constexpr (SAVE) {
                if (lane_o == 0) {        // (overwrites the first dwords of the tile's xyz-encoding slab: probe builds only)
                    unsigned* pr = reinterpret_cast<unsigned*>(tile_base);
                    pr[0] = st.pr_wait; pr[1] = st.pr_bar; pr[2] = st.pr_n;
                    pr[3] = (unsigned)(__builtin_amdgcn_s_memrealtime() - probe_t0);
                }
            }
#endif
#if NERFHIP_CLOCK_PROBE
            if constexpr (F8) {
                if (lane_o == 0) {
                    unsigned* pr = reinterpret_cast<unsigned*>(tile_base + f8_act_scale_off() + 64);
                    pr[0] = (unsigned)(clock64() - probe_c0);
                    pr[1] = (unsigned)(wall_clock64() - probe_w0);
                }
            }
#endif
            return;
        }
    }
}

template <int PREC, int MODE, bool SIGMA_ONLY, int SV>
__global__ __launch_bounds__((KCfg<PREC, (SV != 0)>::NW * 64), (KCfg<PREC, (SV != 0)>::WPS))
void mlp_fwd_kernel(const float* __restrict__ in0, const float* __restrict__ in1, int64_t n, int64_t aux,
                    const uint8_t* __restrict__ packed, float* __restrict__ out, uint8_t* __restrict__ save, FwdZGen zg) {
    __shared__ __attribute__((aligned(1024))) char lds_all[FwdLds<PREC, (SV != 0)>::kBytes];
    mlp_fwd_body<PREC, MODE, SIGMA_ONLY, SV>(lds_all, blockIdx.x, in0, in1, n, aux, packed, out, save, zg);
}

// ---- one kernel instantiation per translation unit ----------------------------------------------------------------
// The 12 instantiations (2 precisions x 2 input modes x {inference, sigma-only, activation-saving}) are fully unrolled
// ~10^4-instruction kernels; compiled in one translation unit they take ~12 minutes of serial hipcc time.
// nerf_pl_amd/build.py therefore compiles mlp_fwd_variant.hip once per (PREC, MODE, VARIANT) with -D flags, in
// parallel; mlp_fwd.hip holds the C ABI and calls these launchers.
template <int PREC, int MODE, bool SIGMA_ONLY, int SV>
int launch_fwd_variant(const float* in0, const float* in1, int64_t n, int64_t aux, const void* packed, float* out, void* save,
                       unsigned blocks, hipStream_t stream, const FwdZGen& zg);

}  // namespace nerfhip
