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
//   bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate, 8 waves (2 per SIMD) = 256 points / workgroup
//   fp32: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), 4 waves (1 per SIMD) = 128 points / workgroup
#pragma once
#include <type_traits>

#include "common.h"
#include "mlp_layout.h"
#include "f8_store.h"

#ifndef NERFHIP_TILE_SCHED_BARRIER
#define NERFHIP_TILE_SCHED_BARRIER 0
#endif
#ifndef NERFHIP_STORE_AUX
#define NERFHIP_STORE_AUX 2     // cache-policy bits of the activation stores: 2 = nt (written once, read by another kernel: -7 %)
#endif
#ifndef NERFHIP_FAST_SINCOS
#define NERFHIP_FAST_SINCOS 1   // bf16 kernels only; the fp32 (parity) kernels always use sincosf
#endif
#ifndef NERFHIP_EXP
#define NERFHIP_EXP 0
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

// Launch geometry.  bf16: 8 waves (2 per SIMD, 256 regs each) = 256 points / workgroup, the two waves of a SIMD
// overlap each other's epilogue VALU / activation stores with MFMA.  fp32: 4 waves (1 per SIMD, 512 regs; an fp32
// slab set is 128 registers).  NERFHIP_SAVE8=0 builds the bf16 training (SAVE) variant with 4 waves as well.
template <int PREC, bool SAVE> struct KCfg {
#ifndef NERFHIP_SAVE8
#define NERFHIP_SAVE8 1    // measured: the 4-wave/512-register SAVE build is bimodal across MI355X boxes (376 us on some,
#endif                     // ~900 us on others, same binary); the 8-wave build is 460-520 us everywhere
    static constexpr int NW = (PREC == NERFHIP_BF16 && (!SAVE || NERFHIP_SAVE8)) ? 8 : 4;
    static constexpr int WPS = (PREC == NERFHIP_BF16 && (!SAVE || NERFHIP_SAVE8)) ? 2 : 1;
    // A-fragment software prefetch depth (bf16): LDS reads issued this many MFMAs ahead of their use, so the
    // ~100-cycle ds_read latency is not exposed once per 32-cycle MFMA
    // (measured at 1024x192: inference forward 197 us with depth 2 vs 204 with depth 1; the 8-wave SAVE variant the
    //  other way round, 238 vs 257 us — its registers are better spent elsewhere)
    static constexpr int PF = (PREC != NERFHIP_BF16) ? 1 : (SAVE ? (NERFHIP_SAVE8 ? 1 : 8) : NERFHIP_PF2);
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

template <int PREC, int NCH, bool COUNT_STORES = false>
struct WeightStream {
    static constexpr int NW = KCfg<PREC, COUNT_STORES>::NW;    // COUNT_STORES == SAVE variant
    static constexpr int LPW = kChunkPieces / NW;   // DMA instructions per wave per chunk
    const uint8_t* gsrc;     // packed + lane*16
    unsigned lds_base;       // LDS byte address of the ring
    int wave;                // wave index in the workgroup (SGPR)
    int pending;             // vector-memory STORE instructions issued since the last boundary (SAVE variant).
                             // Straight-line code: the optimiser folds this to a constant at every boundary.

    __device__ __forceinline__ void issue_piece(int c, int k) const {      // k-th of this wave's LPW pieces of chunk c
        const int piece = wave + k * NW;
        glds16(gsrc + ((size_t)c * kChunkPieces + piece) * kPieceBytes,
               lds_base + (unsigned)((c % kSlots) * kChunkBytes + piece * kPieceBytes));
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
            const int n = (c + 1 < NCH ? LPW : 0) + pending;
            pending = 0;
            wait_barrier(n);
        } else if (c + 1 < NCH) {
            if (LPW == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else          asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (c + 2 < NCH) issue_chunk(c + 2);
    }
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
template <int PREC, int NCH, bool CS>
__device__ __forceinline__ void save_gates(WeightStream<PREC, NCH, CS>& st, uint8_t* tile_ptr, int gate_off, int piece,
                                           const u32x4& g, int lane) {
    // section offset goes into the descriptor BASE (SALU), soffset = 0 (gfx950 store-data hazard, see save_slabs)
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_ptr + gate_off + piece * kPieceBytes, 0,
                                                                  kPieceBytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(g, rs, (unsigned)lane * 16u, 0, 0);
    st.pending += 1;
}
template <int PREC, int NCH, typename Slab, bool CS>
__device__ __forceinline__ void save_slabs(WeightStream<PREC, NCH, CS>& st, uint8_t* tile_ptr, int sec,
                                           const Slab* slabs, int n, int lane) {
    // One descriptor per call with the section offset folded into its (wave-uniform, SALU-computed) base: every store
    // then uses the SAME voffset VGPR (lane * sizeof(Slab)) and a small immediate.  (Folding the section offset into
    // the voffset instead made hipcc hoist ~40 distinct per-lane offset VGPRs to the top of the kernel: +80 live
    // registers, spills.)
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_ptr + (size_t)sec * 64 * sizeof(Slab), 0,
                                                                  (int)(n * 64 * sizeof(Slab)), 0x00020000);
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(Slab);
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const u32x4* src = reinterpret_cast<const u32x4*>(&slabs[i]);
#pragma unroll
        for (int q = 0; q < (int)(sizeof(Slab) / 16); ++q) {
            __builtin_amdgcn_raw_buffer_store_b128(src[q], rs, voff + (unsigned)(i * 64 * sizeof(Slab)) + 16 * q, 0, NERFHIP_STORE_AUX);
            st.pending += 1;
        }
    }
}

__device__ __forceinline__ float slab_absmax(const bf16x8& s) {
    float m = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)s[j]));
    return m;
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

// ---- one layer, OUTPUT-TILE-MAJOR: for each 32-row output tile t: acc = bias; acc += W_frag(t,ks) * B[ks] over all
// input slabs ks; then that tile's epilogue (activation, pack to the next layer's B slabs 2t and 2t+1, ReLU gate
// bits, activation stores) runs while the matrix pipe already works on tile t+1 (other accumulator).  The
// epilogue VALU is thereby spread over the layer in 1/NT portions instead of one block at the layer end, where —
// all waves being chunk-synchronised by the weight ring's barriers — it used to stall every SIMD's MFMA pipe at once.
// Weights are packed in the same (t, ks) order (mlp_pack.hip); fragment i = t*NKS + ks = piece G0 + 1 + i*PPF.
//   out != nullptr : `out[2t], out[2t+1]` receive the activated slabs;   heads (NT == 1) return the raw tile in *raw.
// fp8 storage (SV == 2): a layer stores its INPUT slabs (`chain` = the previous layer's activations, section `in_sec`, live in
// registers for the whole layer) instead of its outputs — pair by pair in its tile epilogues, under the e8m0 scale byte
// `in_sb` the previous layer computed with ONE wave reduction at its end — and returns the scale byte of its own output
// (the per-lane max is folded into the epilogues: 8 v_max3 per tile).  The conversions therefore depend on nothing the
// tile just computed: no reduction latency sits in any tile epilogue (a per-tile scale made this kernel 60 % slower).
template <int PREC, int L, int NCH, int NT, bool RELU, int SV, typename Slab>
__device__ __forceinline__ int run_layer(WeightStream<PREC, NCH, (SV != 0)>& st, const char* smem_lane, char* bias_priv,
                                         const char* enc_lds, const Slab* chain, Slab* out, f32x16* raw,
                                         uint8_t* rsrc, int act_sec, int gate_piece, int lane, int in_sec = -1, int in_sb = 127) {
    constexpr bool SAVE = SV != 0, F8 = SV == 2;
    static_assert(!F8 || PREC == NERFHIP_BF16, "fp8 storage is a bf16-compute mode");
    lane = SAVE ? fresh_lane() : lane;
    constexpr Layer ly = kLayers[L];
    static_assert(ly.nt == NT, "tile count mismatch");
    constexpr int G0 = layer_start(L, PREC);
    constexpr int PPF = ppf(PREC);
    constexpr int NKS = ly.enc_slabs + ly.chain_slabs;
    constexpr int N = NT * NKS;
    constexpr int D = (KCfg<PREC, SAVE>::PF < N) ? KCfg<PREC, SAVE>::PF : N;    // A-fragment prefetch depth

    auto piece_off = [](int g) { return ((g / kChunkPieces) % kSlots) * kChunkBytes + (g % kChunkPieces) * kPieceBytes; };
    // A fragment i: bf16 = one 16-B read per lane; fp32 = two (8 x f32)
    auto load_frag = [&](auto ic, Slab& a) {
        constexpr int i = decltype(ic)::value;
        constexpr int g = G0 + 1 + i * PPF;
        st.template at_piece<g>();
        if constexpr (PREC == NERFHIP_BF16) {
            a = *reinterpret_cast<const bf16x8*>(smem_lane + piece_off(g));
        } else {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(smem_lane + piece_off(g));
            st.template at_piece<g + 1>();
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(smem_lane + piece_off(g + 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = a0[j]; a[4 + j] = a1[j]; }
        }
    };

    st.template at_piece<G0>();
    // The layer's bias piece is needed at the start of EVERY output tile, by which time its ring slot may have been
    // refilled (a layer spans up to 5 chunks): copy it once into this wave's private 1 KiB of LDS (same-wave LDS
    // operations execute in order, so no barrier is needed).
    {
        const u32x4 bv = *reinterpret_cast<const u32x4*>(smem_lane + piece_off(G0));
        *reinterpret_cast<u32x4*>(bias_priv + lane * 16) = bv;
    }
    Slab a[D];
    static_for<0, D>([&](auto ic) { load_frag(ic, a[decltype(ic)::value]); });

#ifndef NERFHIP_GATE_DWORD_STORES
#define NERFHIP_GATE_DWORD_STORES 1   // store each 32-bit gate word as soon as its two tiles are done (one live register
#endif                                // instead of four: the activation-saving kernels run at the 256-register limit)
    f32x16 acc[2];
    unsigned gw[4] = {0u, 0u, 0u, 0u};
    float mx = 0.0f;                                   // F8: this lane's max |output| of the layer

    static_for<0, N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int t = frag_tile(i, NT, NKS), ks = frag_slab(i, NT, NKS);
        f32x16& c = acc[t & 1];
        if constexpr (ks == 0) {
            // bias -> accumulator init.  Row of reg r: 32t + (r&3) + 8(r>>2) + 4h  => one f32x4 per (t, r>>2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias_priv + (lane >> 5) * 16 + (32 * t + 8 * q) * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) c[4 * q + k] = b[k];
            }
        }
        // B operand: an input-encoding slab (parked in this wave's LDS stash)
        // or a slab of the previous layer's activations (registers)
        Slab bs;
        if constexpr (ks < ly.enc_slabs)      // (enc_lds = the wave's stash base; the lane offset is added here, from the fresh lane id)
            bs = *reinterpret_cast<const Slab*>(enc_lds + lane * (int)sizeof(Slab) + ks * 64 * (int)sizeof(Slab));
        else bs = chain[ks - ly.enc_slabs];
        if constexpr (PREC == NERFHIP_BF16) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % D], bs, c, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i % D][j], bs[j], c, 0, 0, 0);
        }
        if constexpr (i + D < N) load_frag(std::integral_constant<int, i + D>{}, a[i % D]);

#ifndef NERFHIP_F8_CVT_AT_START
#define NERFHIP_F8_CVT_AT_START 0
#endif
        if constexpr (ks == (NERFHIP_F8_CVT_AT_START ? 0 : NKS - 1)) {
            if constexpr (F8 && PREC == NERFHIP_BF16) {
                // this tile's share of the layer's INPUT pairs (chain_slabs / 2 pairs spread over the NT tiles, issued in
                // bursts of NERFHIP_F8_BURST tiles' worth: consecutive pairs are contiguous KiBs of HBM)
#ifndef NERFHIP_F8_BURST
#define NERFHIP_F8_BURST 1
#endif
                constexpr int NP = ly.chain_slabs / 2, BU = (NERFHIP_F8_BURST < NT) ? NERFHIP_F8_BURST : NT;
                constexpr int q0 = (t % BU == 0) ? t * NP / NT : 0, q1 = (t % BU == 0) ? (t + BU) * NP / NT : 0;
                if (in_sec >= 0) {
#pragma unroll
                    for (int q = q0; q < q1; ++q)
                        save_pair_f8(st.pending, rsrc, in_sec / 2 + q, chain[2 * q], chain[2 * q + 1], in_sb, lane);
                }
            }
        }
        if constexpr (ks == NKS - 1) {                       // ---- epilogue of tile t ----
            if constexpr (NT == 1) {                         // heads: hand the raw tile back
                *raw = c;
            } else {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = c[8 * sl + j];
                        v[j] = RELU ? fmaxf(x, 0.0f) : x;
#if NERFHIP_F8EXP != 3
                        if (F8) mx = fmaxf(mx, RELU ? v[j] : fabsf(v[j]));
#endif
                        if (RELU && SAVE && NERFHIP_F8EXP != 4) {
                            // ReLU gate of value idx = 8*(2t+sl) + j -> word idx>>5, bit 31-(idx&31).  Pure VALU (no
                            // v_cmp: 128 live SGPR lane masks per layer spill): relu(x) is +-0 or positive, so bit 31 of
                            // (bits + 0x7fffffff) is [x > 0]; v_alignbit pushes it into the word.
                            const int idx = 8 * (2 * t + sl) + j;
                            const unsigned sb = __float_as_uint(v[j]) + 0x7fffffffu;
                            gw[idx >> 5] = __builtin_amdgcn_alignbit(gw[idx >> 5], sb, 31);   // (gw << 1) | (sb >> 31)
                        }
                    }
                    make_slab(out[2 * t + sl], v);
                }
                if constexpr (SAVE && !F8) save_slabs(st, rsrc, act_sec + 2 * t, &out[2 * t], 2, lane);
                if constexpr (SAVE && RELU && NERFHIP_GATE_DWORD_STORES && (t & 1) == 1) {
                    __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
                        rsrc + (F8 ? f8_act_gate_off() : act_mask_off(PREC)) + gate_piece * kPieceBytes, 0, kPieceBytes, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b32(gw[t >> 1], grs, (unsigned)lane * 16u + 4u * (t >> 1), 0, 0);
                    st.pending += 1;
                }
            }
#if NERFHIP_TILE_SCHED_BARRIER
            __builtin_amdgcn_sched_barrier(0);     // keep the scheduler from stretching live ranges across tiles
#endif
        }
    });
    if constexpr (SAVE && RELU && NT != 1 && !NERFHIP_GATE_DWORD_STORES) {
        u32x4 g;
        g[0] = gw[0]; g[1] = gw[1]; g[2] = gw[2]; g[3] = gw[3];
        save_gates(st, rsrc, F8 ? f8_act_gate_off() : act_mask_off(PREC), gate_piece, g, lane);
    }
    if constexpr (F8 && NT != 1) {
        // scale of this layer's OUTPUT section (one reduction per layer), recorded in the tile's scale table
        const int sb = f8_block_scale(mx);
        save_scale_f8(st.pending, rsrc, f8_act_scale_off(), f8_x_section(act_sec), sb, lane);
        return sb;
    }
    return 127;
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

// SV: 0 = inference, 1 = save activations in the compute precision, 2 = save them as block-scaled e4m3 (bf16 compute)
template <int PREC, int MODE, bool SIGMA_ONLY, int SV>
__global__ __launch_bounds__((KCfg<PREC, (SV != 0)>::NW * 64), (KCfg<PREC, (SV != 0)>::WPS))
void mlp_fwd_kernel(const float* __restrict__ in0, const float* __restrict__ in1, int64_t n, int64_t aux,
                    const uint8_t* __restrict__ packed, float* __restrict__ out, uint8_t* __restrict__ save) {
    using Slab = typename PrecTraits<PREC>::Slab;
    constexpr bool SAVE = SV != 0, F8 = SV == 2;
    constexpr int NW = KCfg<PREC, SAVE>::NW;
    constexpr int NCH = SIGMA_ONLY ? chunks_upto_layer(kSigmaLayer + 1, PREC) : chunks_upto_layer(kNumLayers, PREC);
    // LDS: weight ring | per-wave bias copy (1 KiB) | per-wave input-encoding stash (6 slabs: the encodings are
    // needed only by layers 0, 4 (xyz) and 10 (dir); parking them in LDS frees 24 (bf16) / 48 (fp32) registers)
    constexpr int kEncStash = (kXyzSlabs + kDirSlabs) * 64 * (int)sizeof(Slab);
    __shared__ __attribute__((aligned(1024))) char ring[kSlots * kChunkBytes + NW * kPieceBytes + NW * kEncStash];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int h = lane >> 5;
    const int64_t p = (int64_t)blockIdx.x * (32 * NW) + wave * 32 + (lane & 31);
    const bool valid = p < n;
    const int64_t pc = valid ? p : n - 1;

    // ---- raw inputs (ordinary loads, issued before the DMA stream starts) ----
    float xyz[3] = {0.f, 0.f, 0.f}, dir[3] = {0.f, 0.f, 0.f};
    const float* row = nullptr;
    if (MODE == MODE_RAYS) {
        const int64_t S = aux;
        const int64_t ray = pc / S;
        const float zv = in1[pc];
        const float* rp = in0 + ray * 8;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dir[c] = rp[3 + c];
            xyz[c] = nh_add(rp[c], nh_mul(dir[c], zv));   // o + d*z   rendering.py:206-207
        }
    } else {
        row = in0 + pc * aux;
    }

    WeightStream<PREC, NCH, SAVE> st;
    st.gsrc = packed + lane * 16;
    st.lds_base = (unsigned)(uintptr_t)ring;
    st.wave = wave;
    st.pending = 0;
    // wave-uniform base of this wave's activation block (SAVE); the store helpers derive descriptors from it
    uint8_t* tile_base = SAVE ? save + ((size_t)blockIdx.x * NW + wave) * (F8 ? f8_act_tile_bytes() : act_tile_bytes(PREC))
                              : (uint8_t*)nullptr;

    st.issue_chunk(0);
    if (NCH > 1) st.issue_chunk(1);

    const char* smem_lane = ring + lane * 16;
    char* smem_half = ring + kSlots * kChunkBytes + wave * kPieceBytes;    // this wave's private bias copy (run_layer)

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
        save_slabs(st, tile_base, kActEncX, encx, kXyzSlabs, lane);
        save_slabs(st, tile_base, kActEncD, encd, kDirSlabs, lane);
    }
    char* enc_x = ring + kSlots * kChunkBytes + NW * kPieceBytes + wave * kEncStash;      // wave-uniform stash bases
    char* enc_d = enc_x + kXyzSlabs * 64 * (int)sizeof(Slab);
#pragma unroll
    for (int k = 0; k < kXyzSlabs; ++k) *reinterpret_cast<Slab*>(enc_x + lane * (int)sizeof(Slab) + k * 64 * (int)sizeof(Slab)) = encx[k];
    if (!SIGMA_ONLY) {
#pragma unroll
        for (int k = 0; k < kDirSlabs; ++k) *reinterpret_cast<Slab*>(enc_d + lane * (int)sizeof(Slab) + k * 64 * (int)sizeof(Slab)) = encd[k];
    }
    // activations ping-pong between two register slab sets (a layer reads one while its tiles fill the other)
    Slab ha[16], hb[16];
    f32x16 raw;
    int sb = 127;       // e8m0 scale byte of the section produced by the previous layer (F8)
#define NH_LAYER(L, ENC, IN, OUT)                                                                               \
    sb = run_layer<PREC, L, NCH, 8, true, SV>(st, smem_lane, smem_half, ENC, IN, OUT, (f32x16*)nullptr, tile_base, \
                                              act_h(L + 1), mask_piece_h(L + 1), lane, (L) == 0 ? -1 : act_h(L), sb);
    NH_LAYER(0, enc_x, (const Slab*)nullptr, ha)
    NH_LAYER(1, (const char*)nullptr, ha, hb)
    NH_LAYER(2, (const char*)nullptr, hb, ha)
    NH_LAYER(3, (const char*)nullptr, ha, hb)
    NH_LAYER(4, enc_x, hb, ha)
    NH_LAYER(5, (const char*)nullptr, ha, hb)
    NH_LAYER(6, (const char*)nullptr, hb, ha)
    NH_LAYER(7, (const char*)nullptr, ha, hb)              // h8 -> hb
#undef NH_LAYER

    run_layer<PREC, 8, NCH, 1, false, SV>(st, smem_lane, smem_half, (const char*)nullptr, hb, (Slab*)nullptr, &raw,
                                          tile_base, 0, 0, lane);
    const float sigma = raw[0];                              // row 0 lives in reg 0 of the h=0 lanes

    if (SIGMA_ONLY) {
        // (output index from a recomputed lane id: otherwise the 64-bit address computed in the prologue is kept live —
        //  i.e. spilled to scratch — across the whole network)
        const int lane_o = fresh_lane_opaque();
        const int64_t po = (int64_t)blockIdx.x * (32 * NW) + wave * 32 + (lane_o & 31);
        if (po < n && (lane_o >> 5) == 0) out[po] = sigma;  // (n,1)   nerf.py:112-114
        return;
    } else {
        // xyz_encoding_final: no activation (nerf.py:116) -> ha   (F8: stores its input h8)
        sb = run_layer<PREC, 9, NCH, 8, false, SV>(st, smem_lane, smem_half, (const char*)nullptr, hb, ha, (f32x16*)nullptr,
                                                   tile_base, kActFeat, 0, lane, act_h(8), sb);
        // dir_encoding: relu(W [feat | dir])  (nerf.py:118-119) -> hb[0..7]   (F8: stores its input feat)
        sb = run_layer<PREC, 10, NCH, 4, true, SV>(st, smem_lane, smem_half, enc_d, ha, hb, (f32x16*)nullptr, tile_base, kActT,
                                                   kMaskPieceT, lane, kActFeat, sb);
        run_layer<PREC, 11, NCH, 1, false, SV>(st, smem_lane, smem_half, (const char*)nullptr, hb, (Slab*)nullptr, &raw,
                                               tile_base, 0, 0, lane, kActT, sb);                       // (F8: stores its input t)
        const int lane_o = fresh_lane_opaque();
        const int64_t po = (int64_t)blockIdx.x * (32 * NW) + wave * 32 + (lane_o & 31);
        if (po < n && (lane_o >> 5) == 0) {
            float4 o;
            o.x = 1.0f / (1.0f + expf(-raw[0]));            // sigmoid   nerf.py:79-81
            o.y = 1.0f / (1.0f + expf(-raw[1]));
            o.z = 1.0f / (1.0f + expf(-raw[2]));
            o.w = sigma;                                     // cat([rgb, sigma])   nerf.py:122
            reinterpret_cast<float4*>(out)[po] = o;
        }
    }
}

// ---- one kernel instantiation per translation unit ----------------------------------------------------------------
// The 12 instantiations (2 precisions x 2 input modes x {inference, sigma-only, activation-saving}) are fully unrolled
// ~10^4-instruction kernels; compiled in one translation unit they take ~12 minutes of serial hipcc time.
// nerf_pl_amd/build.py therefore compiles mlp_fwd_variant.hip once per (PREC, MODE, VARIANT) with -D flags, in
// parallel; mlp_fwd.hip holds the C ABI and calls these launchers.
template <int PREC, int MODE, bool SIGMA_ONLY, int SV>
int launch_fwd_variant(const float* in0, const float* in1, int64_t n, int64_t aux, const void* packed, float* out, void* save,
                       unsigned blocks, hipStream_t stream);

}  // namespace nerfhip
