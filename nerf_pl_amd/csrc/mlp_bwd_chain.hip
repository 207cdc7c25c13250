// K2b, phase A: the backward chain of the fused NeRF MLP (autograd mirror of reference models/nerf.py:100-124 as driven by
// train.py:103-117 `loss.backward()`): per 32-point wave tile, the same register-resident chain as the forward, run in reverse
// with W^T streamed through the LDS ring:  g_h(l-1) = W_l^T g_a(l),  g_a = g_h * relu'(h)  (gates read from the words the
// forward saved).  Emits every dL/d(pre-activation) as slabs in the forward's fragment order for the weight-gradient GEMM
// (mlp_bwd.hip).  MFMA-bound, ~0.93x the forward's MFMA count.
// Its own translation unit: built with -amdgpu-sched-strategy=max-memory-clause (nerf_pl_amd/build.py), under which the
// fp8-storage variant fits the 256-register budget of its 2-waves-per-SIMD launch bounds without spilling.
#include <type_traits>

#include "common.h"
#include "mlp_layout.h"
#include "f8_store.h"
#include "mlp_bwd_chain.h"

#ifndef NERFHIP_STORE_AUX
#define NERFHIP_STORE_AUX 2  // cache-policy bits of the dY stores: 2 = nt (-7 %; whole training step 1.65 -> 1.51 ms)
#endif
#ifndef NERFHIP_CHAIN_PK_GATE
#define NERFHIP_CHAIN_PK_GATE 1   // bf16 chain: ReLU gates applied to the PACKED bf16 pairs (3 packed ops per pair instead of 2 per value)
#endif
#ifndef NERFHIP_DMA_SADDR
#define NERFHIP_DMA_SADDR 1       // weight-stream DMAs address as SGPR base + one constant per-lane VGPR offset (no per-piece VALU address)
#endif

namespace nerfhip {
using namespace mlp;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PREC> struct BwdTraits;
template <> struct BwdTraits<NERFHIP_BF16> {
    using Slab = bf16x8;
    static constexpr int NW = 8, WPS = 2;
};
template <> struct BwdTraits<NERFHIP_F32> {
    using Slab = f32x8;
    static constexpr int NW = 4, WPS = 1;
};

__device__ __forceinline__ void mk_slab(bf16x8& s, const float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = (__bf16)v[j];
}
__device__ __forceinline__ void mk_slab(f32x8& s, const float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = v[j];
}
__device__ __forceinline__ float slab_absmax8(const bf16x8& s) {
    float m = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)s[j]));
    return m;
}
__device__ __forceinline__ float slab_absmax8(const f32x8& s) { return 0.0f; }   // (fp8 storage is bf16-only; keeps templates uniform)

__device__ __forceinline__ void glds16b(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
// 16 bytes per lane, global -> LDS, source = wave-uniform base (SGPR pair) + per-lane byte offset `voff`
__device__ __forceinline__ void glds16b_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// ================================================================================================
// Phase A: backward chain
// ================================================================================================
// W^T fragments read this many MFMA steps ahead through an explicit register ring (bf16 compute); 0: no ring, hipcc reads each
// fragment right before its MFMA.  Measured (profiles/README.md round 3, same-call A/B, 1024 x 192): the bf16-storing variant
// 226.9 -> 217.5 us with depth 2 (depth 1: 216-218, depth 3: 219); the e5m2-storing variant sits at its 256-register budget,
// where the ring costs 16-48 spilled registers: 209.3 (none) / 212 (1) / 212 (2) / 222 us (3) — the two waves of a SIMD already
// cover each other's LDS round trips there.
#ifndef NERFHIP_CHAIN_BURST
#define NERFHIP_CHAIN_BURST 2
#endif
#ifndef NERFHIP_CHAIN_DEPTH
#define NERFHIP_CHAIN_DEPTH 2
#endif
#ifndef NERFHIP_CHAIN_DEPTH_F8
#define NERFHIP_CHAIN_DEPTH_F8 0
#endif
template <int PREC>
struct BwdStream {
    static constexpr int NW = BwdTraits<PREC>::NW;
    static constexpr int LPW = kChunkPieces / NW;
    static constexpr int NCH = bwd_chunks(PREC);
    const uint8_t* gsrc;       // packed stream + lane * 16 (per-lane address) | NERFHIP_DMA_SADDR: the stream itself (wave-uniform)
    unsigned voff;             // NERFHIP_DMA_SADDR: lane * 16
    unsigned lds_base;
    int wave;
    int pending;   // stores issued since the last boundary (constant-folded; see mlp_fwd.hip)
    int pending_prev;
#if NERFHIP_STREAM_PROBE
    unsigned pr_wait = 0, pr_bar = 0, pr_n = 0;
#endif

    __device__ __forceinline__ void issue_chunk(int c) const {
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int piece = wave + i * NW;
#if NERFHIP_DMA_SADDR
            glds16b_s(gsrc + ((size_t)c * kChunkPieces + piece) * kPieceBytes, voff,
                      lds_base + (unsigned)((c % kSlots) * kChunkBytes + piece * kPieceBytes));
#else
            glds16b(gsrc + ((size_t)c * kChunkPieces + piece) * kPieceBytes,
                    lds_base + (unsigned)((c % kSlots) * kChunkBytes + piece * kPieceBytes));
#endif
        }
    }
    __device__ __forceinline__ void boundary(int c) {
        // chunk c's DMAs were issued at boundary c-2: younger than them are the stores of the interval before the previous
        // boundary (pending_prev), chunk c+1's DMAs and the stores since the previous boundary => stores get two chunk
        // intervals to retire before a boundary waits for them
        const int n = (c + 1 < NCH ? LPW : 0) + pending + (NERFHIP_STORE_SLACK ? pending_prev : 0);
        pending_prev = pending;
        pending = 0;
#if NERFHIP_STREAM_PROBE
#define NH_WBAR
        const unsigned t0 = (unsigned)__builtin_amdgcn_s_memrealtime();
#else
#define NH_WBAR "\n\ts_barrier"
#endif
#define NH_WB(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" NH_WBAR ::: "memory"); break;
        switch (n < 0 ? 0 : (n > 48 ? 48 : (n <= 8 ? n : (n & ~3)))) {
            NH_WB(0) NH_WB(1) NH_WB(2) NH_WB(3) NH_WB(4) NH_WB(5) NH_WB(6) NH_WB(7) NH_WB(8)
            NH_WB(12) NH_WB(16) NH_WB(20) NH_WB(24) NH_WB(28) NH_WB(32) NH_WB(36) NH_WB(40) NH_WB(44) NH_WB(48)
            default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" NH_WBAR ::: "memory"); break;
        }
#undef NH_WB
#undef NH_WBAR
#if NERFHIP_STREAM_PROBE
        const unsigned t1 = (unsigned)__builtin_amdgcn_s_memrealtime();
        asm volatile("s_barrier" ::: "memory");
        const unsigned t2 = (unsigned)__builtin_amdgcn_s_memrealtime();
        pr_wait += t1 - t0; pr_bar += t2 - t1; pr_n += 1;
#endif
        if (c + 2 < NCH) issue_chunk(c + 2);
    }
};

template <typename Slab>
__device__ __forceinline__ Slab load_slab(__amdgpu_buffer_rsrc_t rsrc, int sec, int lane) {
    Slab s;
    u32x4* dst = reinterpret_cast<u32x4*>(&s);
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(Slab);
    const unsigned soff = (unsigned)(sec * 64 * sizeof(Slab));
#pragma unroll
    for (int q = 0; q < (int)(sizeof(Slab) / 16); ++q) dst[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16 * q, soff, 0);
    return s;
}
template <int PREC, typename Slab>
__device__ __forceinline__ void store_slab(BwdStream<PREC>& st, __amdgpu_buffer_rsrc_t rsrc, int sec, const Slab& s, int lane) {
    const u32x4* src = reinterpret_cast<const u32x4*>(&s);
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(Slab);
    const unsigned soff = (unsigned)(sec * 64 * sizeof(Slab) * act_il(PREC));        // (slab modes only: pieces IL KiB apart, mlp_layout.h)
#pragma unroll
    for (int q = 0; q < (int)(sizeof(Slab) / 16); ++q) {
        // soffset must stay 0 (offset folded into VOFFSET): gfx950 store-data hazard, see mlp_fwd.hip save_slabs
        __builtin_amdgcn_raw_buffer_store_b128(src[q], rsrc, voff + soff + 16 * q, 0, NERFHIP_STORE_AUX);
        st.pending += 1;
    }
}

// dY pair store / block scale in the configured 8-bit format (f8_store.h: e5m2 by default, see NERFHIP_F8_DY_E5M2)
__device__ __forceinline__ void save_dy_pair(int& pending, uint8_t* dy_tile, int pair, const bf16x8& s0, const bf16x8& s1, int sb, int lane) {
#if NERFHIP_F8_DY_E5M2
    save_pair_bf8(pending, dy_tile, pair, s0, s1, sb, lane);
#else
    save_pair_f8(pending, dy_tile, pair, s0, s1, sb, lane);
#endif
}
__device__ __forceinline__ void save_dy_pair(int&, uint8_t*, int, const f32x8&, const f32x8&, int, int) {}    // (never used: fp8 storage is bf16-only)
__device__ __forceinline__ int dy_block_scale(float lane_max) {
#if NERFHIP_F8_DY_E5M2
    return bf8_block_scale(lane_max);
#else
    return f8_block_scale(lane_max);
#endif
}

// sign-extended one-bit field of a gate word: 0 or ~0.  (As inline asm: written with the builtin or plain C, hipcc turns the
// constant-position extract + AND into v_and + v_cmp + v_cndmask, three VALU per value instead of two.)
template <int BIT>
__device__ __forceinline__ unsigned gate_mask(unsigned word) {
    unsigned m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "n"(BIT));
    return m;
}

template <int B, int E, typename F>
__device__ __forceinline__ void bwd_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        bwd_static_for<B + 1, E>(f);
    }
}

// ---- one backward layer, OUTPUT-TILE-MAJOR: for each 32-row tile t of g_h(l-1) = W_l^T g_a(l):
// 16-17 chained MFMAs over the input slabs, then the tile's epilogue — ReLU gate, bf16 pack into slabs 2t, 2t+1 of the OTHER
// slab set (a layer reads one set while its tiles fill the other), stores — which overlaps the next tile's MFMAs (other
// accumulator) instead of forming one ~420-instruction VALU block per layer during which the matrix pipe idles.
// fp8 storage: like the forward, a layer stores its INPUT section (`gin`, live for the whole layer; `in_pairs` slab pairs
// spread over the tiles, under the scale byte `in_sb` computed by ONE reduction at the end of the layer that produced it)
// and returns the scale byte of its own output.
template <int PREC, int L, int NT, int NKS, bool MASK, bool F8, int IN_PAIRS, typename Slab>
__device__ __forceinline__ int run_bwd_layer_tm(BwdStream<PREC>& st, const unsigned lds_lo, const unsigned lds_hi, const unsigned sig_lds,
                                                const Slab (&gin)[NKS - kBwdLayers[L].sigma_slab], Slab* out,
                                                __amdgpu_buffer_rsrc_t acts, int gate_off, int mask_piece,
                                                __amdgpu_buffer_rsrc_t dys, uint8_t* dy_tile, int dy_sec, int dy_scale_idx, int in_sec,
                                                int in_sb, int lane) {
    constexpr int G0 = bwd_layer_start(L, PREC);
    constexpr int PPF = ppf(PREC);
    static_assert(kBwdLayers[L].nt == NT && kBwdLayers[L].nks == NKS, "bwd layer shape mismatch");
    auto piece_off = [](int g) { return ((g / kChunkPieces) % kSlots) * kChunkBytes + (g % kChunkPieces) * kPieceBytes; };
    // ds_read offsets are 16-bit immediates and the ring is 96 KiB: pieces of its last third are addressed from a second base
    // register (left to itself hipcc materialises one address VGPR per such piece — dozens of live registers)
    // (`lds_hi` is opaque to the optimiser: it would fold the constant back into one address per piece)
    typedef __attribute__((address_space(3))) const char* lds_cptr;
    auto piece_ptr = [&](int g) { return piece_off(g) < 65536 ? (lds_cptr)(lds_lo + (unsigned)piece_off(g)) : (lds_cptr)(lds_hi + (unsigned)(piece_off(g) - 65536)); };
    // B operand of slab step ks: the layer's register slabs; the last slab of the W_c^T + sigma^T layer (the sigma head's
    // gradient) is re-read from the wave's LDS stash for every tile instead of living in four more registers
    constexpr bool SIG = kBwdLayers[L].sigma_slab != 0;
    constexpr int NIN = NKS - (SIG ? 1 : 0);
    auto bslab = [&](int ks) -> Slab {
        if constexpr (SIG) {
            if (ks == NIN) return *reinterpret_cast<__attribute__((address_space(3))) const Slab*>(sig_lds);
        }
        return gin[ks < NIN ? ks : 0];
    };
    u32x4 gates = {0u, 0u, 0u, 0u};
    if (MASK)
        gates = __builtin_amdgcn_raw_buffer_load_b128(acts, (unsigned)lane * 16u, (unsigned)((gate_off + mask_piece * kPieceBytes) * act_il(PREC, F8)), 0);
    float mx = 0.0f;
    f32x16 acc2[2];
    // compile-time loop over the tiles: guarantees static register indexing of the slab arrays (a `#pragma unroll` loop of
    // this size is not always fully unrolled, and one runtime index sends a whole 17-slab array to scratch memory)
    // bf16: the W^T fragments travel through a ring of kChainDepth registers, read kChainDepth MFMA steps ahead of their use —
    // also across the tiles of the layer — and a sched_barrier after every step pins that order.  (Left to itself hipcc reads each
    // fragment into the same four registers right before its MFMA: 480 of the kernel's 716 MFMAs sit behind an
    // `s_waitcnt lgkmcnt(0)` of their own.)
    constexpr int NFR = NT * NKS;
    constexpr int kChainDepth = F8 ? NERFHIP_CHAIN_DEPTH_F8 : NERFHIP_CHAIN_DEPTH;
    [[maybe_unused]] bf16x8 afr[kChainDepth > 0 ? kChainDepth : 1];
    [[maybe_unused]] auto frag_read = [&](auto ic) -> bf16x8 {
        constexpr int g = G0 + decltype(ic)::value * PPF;
        if constexpr (g % kChunkPieces == 0) st.boundary(g / kChunkPieces);
        return *reinterpret_cast<__attribute__((address_space(3))) const bf16x8*>(piece_ptr(g));
    };
    if constexpr (PREC == NERFHIP_BF16 && kChainDepth > 0) {
        bwd_static_for<0, (kChainDepth < NFR ? kChainDepth : NFR)>([&](auto jc) { afr[decltype(jc)::value] = frag_read(jc); });
    }
    bwd_static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        f32x16& acc = acc2[t & 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        if constexpr (PREC == NERFHIP_BF16 && kChainDepth > 0) {
            bwd_static_for<0, NKS>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                constexpr int i = t * NKS + ks;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[i % (kChainDepth > 0 ? kChainDepth : 1)], bslab(ks), acc, 0, 0, 0);
                if constexpr (i + kChainDepth < NFR) afr[i % (kChainDepth > 0 ? kChainDepth : 1)] = frag_read(std::integral_constant<int, i + kChainDepth>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int g = G0 + (t * NKS + ks) * PPF;
            if (g % kChunkPieces == 0) st.boundary(g / kChunkPieces);
            if constexpr (PREC == NERFHIP_BF16) {
                const bf16x8 a = *reinterpret_cast<__attribute__((address_space(3))) const bf16x8*>(piece_ptr(g));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bslab(ks), acc, 0, 0, 0);
            } else {
                const f32x4 a0 = *reinterpret_cast<__attribute__((address_space(3))) const f32x4*>(piece_ptr(g));
                if ((g + 1) % kChunkPieces == 0) st.boundary((g + 1) / kChunkPieces);
                const f32x4 a1 = *reinterpret_cast<__attribute__((address_space(3))) const f32x4*>(piece_ptr(g + 1));
                const Slab bs = bslab(ks);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bs[j], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bs[4 + j], acc, 0, 0, 0);
            }
        }
        }
        if constexpr (F8 && PREC == NERFHIP_BF16) {                  // this tile's share of the INPUT section's pairs
            if constexpr (IN_PAIRS > 0) {
#pragma unroll
                for (int q = 0; q < IN_PAIRS; ++q)
                    if (q >= t * IN_PAIRS / NT && q < (t + 1) * IN_PAIRS / NT)        // folds: t is an unrolled constant
                        save_dy_pair(st.pending, dy_tile, in_sec / 2 + q, gin[2 * q], gin[2 * q + 1], in_sb, lane);
            }
        }
        // ---- epilogue of tile t: g wrt pre-activation = g * relu'(pre-act) (gate bit: mlp_layout.h gate_word / gate_bit) ----
#if NERFHIP_CHAIN_PK_GATE
        if constexpr (PREC == NERFHIP_BF16 && MASK) {
            // the gates on PACKED pairs: dword k of a gate word holds the pair's two bits at bit 15 - k of its half-words, so
            // (half << k) >> 15 (arithmetic) is 0xffff / 0 per half: shift, shift, and = 3 packed operations per pair (2 per
            // VALUE before).  The running maximum is taken over the UNGATED values (one v_max3 per pair): a valid, at most
            // slightly looser bound for the section's scale.
            bwd_static_for<0, 8>([&](auto pc) {
                constexpr int p = decltype(pc)::value;               // pair p: values r = 2p, 2p + 1 -> slab 2t + (p >> 2), dword p & 3
                constexpr int idx = 8 * (2 * t) + 2 * p;
                typedef __attribute__((ext_vector_type(2))) float f32x2v;
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2q;
                typedef __attribute__((ext_vector_type(2))) short s16x2q;
                const f32x2v xv = {acc[2 * p], acc[2 * p + 1]};
                if (F8) mx = fmaxf(fmaxf(mx, fabsf(xv[0])), fabsf(xv[1]));
                const bf16x2q pk = __builtin_convertvector(xv, bf16x2q);
                // (through a scalar copy: __builtin_bit_cast applied directly to the vector ELEMENT reads element 0 whatever the
                // index — hipcc 7.2 — and the gates of tiles 2.. would silently be those of tiles 0, 1)
                const unsigned gword = gates[gate_word(idx)];
                s16x2q gm = __builtin_bit_cast(s16x2q, gword);
                gm = (gm << (short)((idx & 31) >> 1)) >> (short)15;
                const unsigned d = __builtin_bit_cast(unsigned, pk) & __builtin_bit_cast(unsigned, gm);
                const bf16x2q gated = __builtin_bit_cast(bf16x2q, d);
                out[2 * t + (p >> 2)][2 * (p & 3)] = gated[0];
                out[2 * t + (p >> 2)][2 * (p & 3) + 1] = gated[1];
            });
        } else
#endif
        {
        float v[16];
        bwd_static_for<0, 16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;           // slab 2t + (r >> 3), slot r & 7
            constexpr int idx = 8 * (2 * t) + r;
            const float gv = acc[r];
            if constexpr (MASK) v[r] = __uint_as_float(__float_as_uint(gv) & gate_mask<gate_bit(idx)>(gates[gate_word(idx)]));
            else v[r] = gv;
            if (F8) mx = fmaxf(mx, fabsf(v[r]));
        });
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = v[8 * sl + j];
            mk_slab(out[2 * t + sl], v8);
        }
        }
        if constexpr (!F8) {
            // per-layer descriptor: the (possibly runtime, wave-uniform) section offset sits in its SALU-computed base, the
            // per-tile offsets are immediates (soffset stays 0: gfx950 store-data hazard, see store_slab)
            __amdgpu_buffer_rsrc_t dys_l = __builtin_amdgcn_make_buffer_rsrc(dy_tile + (size_t)dy_sec * 64 * sizeof(Slab) * act_il(PREC), 0,
                                                                              (int)(2 * NT * 64 * sizeof(Slab) * act_il(PREC)), 0x00020000);
            // NERFHIP_CHAIN_BURST slabs per run of back-to-back stores (2 = each tile's pair as soon as it is gated; the slabs of a
            // layer stay in registers as the next layer's operands anyway, so holding a run back costs no register)
            constexpr int TB = NERFHIP_CHAIN_BURST / 2 > 0 ? NERFHIP_CHAIN_BURST / 2 : 1;
            if constexpr ((t + 1) % TB == 0 || t == NT - 1) {
                constexpr int t0 = (t / TB) * TB;
#pragma unroll
                for (int sl = 2 * t0; sl < 2 * t + 2; ++sl) store_slab(st, dys_l, sl, out[sl], lane);
            }
        }
    });
    if constexpr (F8 && PREC == NERFHIP_BF16) {
        const int sb = dy_block_scale(mx);
        save_scale_f8(st.pending, dy_tile, f8_dy_scale_off(), dy_scale_idx, sb, lane);
        return sb;
    }
    return 127;
}

template <int PREC, bool F8>
__global__ __launch_bounds__(BwdTraits<PREC>::NW * 64, BwdTraits<PREC>::WPS)
void mlp_bwd_chain_kernel(BwdChainArgs A, const float* __restrict__ g_scale) {
    // ONE launch runs the chains of up to two models (a training step's fine and coarse network): the first A.blocks0 workgroups
    // belong to model 0, the rest to model 1 — wave-uniform selects of the per-model pointers, nothing else changes
    const int mdl = ((int)blockIdx.x >= A.blocks0) ? 1 : 0;
    const unsigned wg = blockIdx.x - (mdl ? (unsigned)A.blocks0 : 0u);
    const float* __restrict__ g_out = mdl ? A.g_out[1] : A.g_out[0];
    const float* __restrict__ out = mdl ? A.out[1] : A.out[0];
    const int64_t n = mdl ? A.n[1] : A.n[0];
    const uint8_t* __restrict__ packed_bwd = mdl ? A.packed_bwd[1] : A.packed_bwd[0];
    const uint8_t* __restrict__ acts_base = mdl ? A.acts[1] : A.acts[0];
    uint8_t* __restrict__ dys_base = mdl ? A.dys[1] : A.dys[0];
    static_assert(!F8 || PREC == NERFHIP_BF16, "fp8 storage is a bf16-compute mode");
    using Slab = typename BwdTraits<PREC>::Slab;
    constexpr int kActTile = F8 ? f8_act_tile_bytes() : act_tile_bytes(PREC);
    constexpr int kGateOff = F8 ? f8_act_gate_off() : act_mask_off(PREC);
    constexpr int kDyTile = F8 ? f8_dy_tile_bytes() : kDySlabs * 64 * (int)sizeof(Slab);
    constexpr int NW = BwdTraits<PREC>::NW;
    __shared__ __attribute__((aligned(1024))) char ring[kSlots * kChunkBytes + NW * 64 * (int)sizeof(Slab)];      // W^T ring | sigma-slab stash
#if NERFHIP_STREAM_PROBE
    const uint64_t probe_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int h = lane >> 5;
    const int64_t tile = (int64_t)wg * NW + wave;
    const int64_t p = tile * 32 + (lane & 31);
    const bool valid = p < n;
    const int64_t pc = valid ? p : n - 1;

    float4 g = reinterpret_cast<const float4*>(g_out)[pc];
    const float4 o = reinterpret_cast<const float4*>(out)[pc];
    if (!valid) g = make_float4(0.f, 0.f, 0.f, 0.f);               // padded points contribute nothing
    if (g_scale) {                                                 // upstream d L / d loss as a device scalar (NULL = 1)
        const float sc = *g_scale;
        g.x *= sc; g.y *= sc; g.z *= sc; g.w *= sc;
    }

    constexpr int IL = act_il(PREC, F8);     // the saved blocks' pieces are IL KiB apart (mlp_layout.h: bf16 slabs 8, otherwise 1)
    __amdgpu_buffer_rsrc_t acts = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(acts_base) + tile_block_off(tile, kActTile, IL), 0, kActTile * IL, 0x00020000);
#ifdef NERFHIP_EXP_TILEWRAP    // timing experiment only (results invalid): every wave stores into one of a few L2-resident tile blocks
    uint8_t* dy_tile = dys_base + tile_block_off(tile & (NERFHIP_EXP_TILEWRAP - 1), kDyTile, IL);
#else
    uint8_t* dy_tile = dys_base + tile_block_off(tile, kDyTile, IL);
#endif
    __amdgpu_buffer_rsrc_t dys = __builtin_amdgcn_make_buffer_rsrc(dy_tile, 0, kDyTile * IL, 0x00020000);


    BwdStream<PREC> st;
#if NERFHIP_DMA_SADDR
    st.gsrc = packed_bwd;
#else
    st.gsrc = packed_bwd + lane * 16;
#endif
    st.voff = (unsigned)lane * 16u;
    st.lds_base = (unsigned)(uintptr_t)ring;
    st.wave = wave;
    st.pending = 0;
    st.pending_prev = 0;
    st.issue_chunk(0);
    st.issue_chunk(1);
    // this lane's 16 bytes of every piece: LDS byte address of the ring's first 64 KiB and (opaque) of the rest
    const unsigned lds_lo = (unsigned)(uintptr_t)ring + (unsigned)lane * 16u;
    unsigned lds_hi = lds_lo + 65536u;
    asm volatile("" : "+v"(lds_hi));
    const unsigned sig_lds = (unsigned)(uintptr_t)ring + (unsigned)(kSlots * kChunkBytes) + (unsigned)((wave * 64 + lane) * (int)sizeof(Slab));

    // d sigmoid: g_a_rgb = g_rgb * rgb * (1 - rgb)      (nerf.py:79-81, 120);  sigma is linear (nerf.py:112)
    float v[8];
    Slab zero_slab;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    mk_slab(zero_slab, v);
    Slab g_rgb, g_sig;
    {
        const float ga[3] = {g.x * o.x * (1.0f - o.x), g.y * o.y * (1.0f - o.y), g.z * o.z * (1.0f - o.z)};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (h == 0 && j < 3) ? ga[j < 3 ? j : 0] : 0.0f;
        mk_slab(g_rgb, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (h == 0 && j == 0) ? g.w : 0.0f;
        mk_slab(g_sig, v);
    }
    if constexpr (F8) {
        if constexpr (PREC == NERFHIP_BF16) {
#if NERFHIP_F8_DY_E5M2
            const int sb_rgb = bf8_block_scale(slab_absmax8(g_rgb)), sb_sig = bf8_block_scale(slab_absmax8(g_sig));
            save_pair_bf8(st.pending, dy_tile, kDyRgb / 2, g_rgb, zero_slab, sb_rgb, lane);
            save_pair_bf8(st.pending, dy_tile, kDySigma / 2, g_sig, zero_slab, sb_sig, lane);
#else
            const int sb_rgb = f8_block_scale(slab_absmax8(g_rgb)), sb_sig = f8_block_scale(slab_absmax8(g_sig));
            save_pair_f8(st.pending, dy_tile, kDyRgb / 2, g_rgb, zero_slab, sb_rgb, lane);
            save_pair_f8(st.pending, dy_tile, kDySigma / 2, g_sig, zero_slab, sb_sig, lane);
#endif
            save_scale_f8(st.pending, dy_tile, f8_dy_scale_off(), f8_dy_section(kDyRgb), sb_rgb, lane);
            save_scale_f8(st.pending, dy_tile, f8_dy_scale_off(), f8_dy_section(kDySigma), sb_sig, lane);
        }
    } else {
        store_slab(st, dys, kDyRgb, g_rgb, lane);
        store_slab(st, dys, kDyRgb + 1, zero_slab, lane);
        store_slab(st, dys, kDySigma, g_sig, lane);
        store_slab(st, dys, kDySigma + 1, zero_slab, lane);
    }

    // scale bytes of the sections produced so far (F8); rgb / sigma pairs were stored above
    Slab gd[8];
    Slab ga[16], gb[16];
    // rgb^T : g_t = W_rgb^T g_a_rgb ; mask with t = relu(dir pre-act)  -> dY_dir in gd
    Slab g_in0[1] = {g_rgb};
    int sb = run_bwd_layer_tm<PREC, 0, 4, 1, true, F8, 0>(st, lds_lo, lds_hi, sig_lds, g_in0, gd, acts, kGateOff, kMaskPieceT, dys, dy_tile, kDyDir,
                                                          f8_dy_section(kDyDir), -1, 127, lane);
    // (the sigma head's slab is the 9th K slab of the next layer: parked in LDS, see run_bwd_layer_tm)
    *reinterpret_cast<__attribute__((address_space(3))) Slab*>(sig_lds) = g_sig;
    // W_c^T + sigma^T : g_h8 = (W_dir[:, :256] W_final)^T g_a_dir + W_sigma^T g_sigma ; mask with h8  -> dY_8 in gb
    // (xyz_encoding_final has no activation: folded, mlp_layout.h kLayers / kBwdLayers; F8: stores its input dY_dir)
    static_assert(kBwdLayerFold == 1, "layer sequence below");
    sb = run_bwd_layer_tm<PREC, 1, 8, 9, true, F8, 4>(st, lds_lo, lds_hi, sig_lds, gd, gb, acts, kGateOff, mask_piece_h(8), dys, dy_tile, dy_h(8),
                                                      f8_dy_section(dy_h(8)), kDyDir, sb, lane);
    // ---- layers 2..7 (L8^T .. L3^T): ONE copy of the code of three layers, run twice.  Fully unrolled, the chain is 67-80 KB of
    // straight-line code against a 64 KiB instruction cache (profiles/archive/r02_slowbox_diagnosis.txt: on some MI355X boxes a kernel
    // over that size runs 1.5x slower for its instruction fetches alone).  A 256 x 256 layer is 4 chunks of the W^T stream, so
    // three layers later the stream is at the same piece offset within a chunk AND in the same ring slot (12 chunks = 0 mod 3):
    // the second pass differs only in wave-uniform values — the stream pointer (+12 chunks), the gate piece (-3), the dY
    // sections (+48 slabs) and scale dwords (+3).  Three layers flip the ga / gb ping-pong, so each pass ends by copying its
    // result back (64 register moves per 384 MFMAs).
    static_assert(bwd_layer_pieces(2, PREC) * 3 % (kChunkPieces * kSlots) == 0, "three looped layers = a whole number of ring turns");
#define NH_BWD(L, IN, OUT, D)                                                                                                \
    sb = run_bwd_layer_tm<PREC, L, 8, 16, true, F8, 8>(st, lds_lo, lds_hi, sig_lds, reinterpret_cast<const Slab(&)[16]>(IN), OUT, acts, kGateOff, \
                                                       mask_piece_h(9 - L) - (D), dys, dy_tile, dy_h(9 - L) + 16 * (D),              \
                                                       f8_dy_section(dy_h(9 - L)) + (D), dy_h(10 - L) + 16 * (D), sb, lane);
    if constexpr (PREC != NERFHIP_BF16) {
        // exact-fp32 variant (the parity configuration, one wave per SIMD): fully unrolled — its code is far beyond the instruction
        // cache either way (186 vs 124 KB) and the loop costs it 27 spilled registers
        NH_BWD(2, gb, ga, 0) NH_BWD(3, ga, gb, 0) NH_BWD(4, gb, ga, 0) NH_BWD(5, ga, gb, 0) NH_BWD(6, gb, ga, 0) NH_BWD(7, ga, gb, 0)
    } else {
        const uint8_t* const gsrc0 = st.gsrc;
        int n_pass;
        asm volatile("s_mov_b32 %0, 2" : "=s"(n_pass));          // opaque trip count: the loop must stay a loop
#pragma clang loop unroll(disable)
        for (int pass = 0; pass < n_pass; ++pass) {
            // (the store counters restart from 0 in every pass: under-counting only over-waits at the first boundaries)
            st.pending = 0;
            st.pending_prev = 0;
            const int d = 3 * pass;                              // layers the pass is ahead of the code's constants
            NH_BWD(2, gb, ga, d) NH_BWD(3, ga, gb, d) NH_BWD(4, gb, ga, d)
#pragma unroll
            for (int i = 0; i < 16; ++i) gb[i] = ga[i];
            st.gsrc += (size_t)(3 * bwd_layer_pieces(2, PREC)) * kPieceBytes;
        }
        st.pending = 0;
        st.pending_prev = 0;
        st.gsrc = gsrc0;
    }
    NH_BWD(8, gb, ga, 0)
#undef NH_BWD
    if constexpr (F8 && PREC == NERFHIP_BF16) {              // the last section (dY_1) has no consuming layer: flush it
#pragma unroll
        for (int q = 0; q < 8; ++q) save_dy_pair(st.pending, dy_tile, dy_h(1) / 2 + q, ga[2 * q], ga[2 * q + 1], sb, lane);
    }
#if NERFHIP_STREAM_PROBE
    if (lane == 0) {                  // (overwrites the first dwords of the tile's dY block: probe builds only)
        unsigned* pr = reinterpret_cast<unsigned*>(dy_tile);
        pr[0] = st.pr_wait; pr[1] = st.pr_bar; pr[2] = st.pr_n;
        pr[3] = (unsigned)(__builtin_amdgcn_s_memrealtime() - probe_t0);
    }
#endif
}


}  // namespace nerfhip

namespace nerfhip {
void launch_bwd_chain(int n_models, const float* const* g_out, const float* g_scale, const float* const* out, const int64_t* n,
                      const void* const* packed_bwd, const void* const* acts, void* const* dys, int dtype, const int64_t* tiles,
                      hipStream_t s) {
    const int wpw = (dtype == NERFHIP_F32) ? 4 : 8;                 // waves (32-point tiles) per workgroup
    BwdChainArgs A;
    unsigned blocks = 0;
    for (int m = 0; m < 2; ++m) {
        const int mm = m < n_models ? m : 0;
        A.g_out[m] = g_out[mm]; A.out[m] = out[mm]; A.n[m] = n[mm];
        A.packed_bwd[m] = (const uint8_t*)packed_bwd[mm]; A.acts[m] = (const uint8_t*)acts[mm]; A.dys[m] = (uint8_t*)dys[mm];
        if (m == 0) A.blocks0 = (int)(tiles[0] / wpw);
        if (m < n_models) blocks += (unsigned)(tiles[m] / wpw);
    }
    if (dtype == NERFHIP_BF16_F8)
        hipLaunchKernelGGL((mlp_bwd_chain_kernel<NERFHIP_BF16, true>), dim3(blocks), dim3(512), 0, s, A, g_scale);
    else if (dtype == NERFHIP_BF16)
        hipLaunchKernelGGL((mlp_bwd_chain_kernel<NERFHIP_BF16, false>), dim3(blocks), dim3(512), 0, s, A, g_scale);
    else
        hipLaunchKernelGGL((mlp_bwd_chain_kernel<NERFHIP_F32, false>), dim3(blocks), dim3(256), 0, s, A, g_scale);
}
}  // namespace nerfhip
