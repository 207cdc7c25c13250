// fp8 (OCP e4m3, block-scaled) storage of the tensors saved for the weight-gradient GEMM — shared by the activation-saving
// forward (mlp_fwd_kernel.h) and the backward chain (mlp_bwd.hip).  Format: mlp_layout.h "fp8 storage of the saved tensors".
#pragma once
#include "common.h"
#include "mlp_layout.h"

#ifndef NERFHIP_STORE_AUX
#define NERFHIP_STORE_AUX 2     // cache-policy bits of the write-once stores: 2 = nt
#endif
#ifndef NERFHIP_F8EXP
#define NERFHIP_F8EXP 0         // timing experiments only (results invalid): 1 = convert but do not store, 2 = store without converting
#endif

#ifndef NERFHIP_STORE_SLACK
#define NERFHIP_STORE_SLACK 1   // weight-ring boundaries let the stores of the last TWO chunk intervals stay in flight (0: one)
#endif

namespace nerfhip {
using namespace mlp;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// ---- training, fp8 storage (NERFHIP_BF16_F8, mlp_layout.h "fp8 storage of the saved tensors") -------------------------
// max over the wave of a non-negative fp32 bit pattern (as unsigned): 6 DPP steps, result in an SGPR
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define NH_DPP_MAX(CTRL, ROWMASK)                                                                  \
    {                                                                                              \
        const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, true); \
        v = v > t ? v : t;                                                                         \
    }
    NH_DPP_MAX(0x111, 0xf)   // row_shr:1
    NH_DPP_MAX(0x112, 0xf)   // row_shr:2
    NH_DPP_MAX(0x114, 0xf)   // row_shr:4
    NH_DPP_MAX(0x118, 0xf)   // row_shr:8   -> lane 15 of every row holds the row maximum
    NH_DPP_MAX(0x142, 0xa)   // row_bcast:15 -> rows 1 and 3 fold in the previous row
    NH_DPP_MAX(0x143, 0xc)   // row_bcast:31 -> rows 2 and 3 fold in lane 31
#undef NH_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// e8m0 scale byte of a block whose largest magnitude has the fp32 bit pattern `maxbits`: E = max(Emax - 7, 1), so that
// |x| / 2^(E-127) < 2^8 (e4m3 tops out at 448 and the conversion does not saturate)
__device__ __forceinline__ int f8_scale_byte(unsigned maxbits) {
    const int e = (int)((maxbits >> 23) & 0xffu) - 7;
    return e < 1 ? 1 : e;
}
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
// 8 bf16 (one slab of one lane) -> 8 e4m3 bytes (2 dwords), x / scale, round to nearest even
__device__ __forceinline__ void slab_to_f8(const bf16x8& s, float scale, unsigned& d0, unsigned& d1) {
    union { bf16x8 v; bf16x2v p[4]; } u;
    u.v = s;
    union { s16x2 h; unsigned w; } a, b;
    a.w = 0u;
    b.w = 0u;
    a.h = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(a.h, u.p[0], scale, false);
    a.h = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(a.h, u.p[1], scale, true);
    b.h = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(b.h, u.p[2], scale, false);
    b.h = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(b.h, u.p[3], scale, true);
    d0 = a.w;
    d1 = b.w;
}
// slab pair (slabs 2*pair, 2*pair+1) as one e4m3 piece under the (wave-uniform) scale byte `sb`
__device__ __forceinline__ void save_pair_f8(int& pending, uint8_t* tile_ptr, int pair, const bf16x8& s0, const bf16x8& s1, int sb,
                                             int lane) {
    const float scale = __uint_as_float((unsigned)sb << 23);
    u32x4 pk;
    unsigned d0, d1;
    slab_to_f8(s0, scale, d0, d1);
    pk[0] = d0; pk[1] = d1;
    slab_to_f8(s1, scale, d0, d1);
    pk[2] = d0; pk[3] = d1;
#if NERFHIP_F8EXP == 1
    asm volatile("" ::"v"(pk));
#else
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_ptr + (size_t)pair * kPieceBytes, 0, kPieceBytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(pk, rs, (unsigned)lane * 16u, 0, NERFHIP_STORE_AUX);
    pending += 1;
#endif
}
// one scale dword (activations: one per (wave tile, section)), written by lane 0
__device__ __forceinline__ void save_scale_f8(int& pending, uint8_t* tile_ptr, int scale_off, int index, int sb, int lane) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_ptr + scale_off + 4 * index, 0, 4, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)sb, rs, (unsigned)lane * 4u, 0, 0);      // lanes 1..63: out of bounds, dropped
    pending += 1;
}
// ---- e5m2 ("bf8") variant for dY: 5 exponent bits keep the heavy tail of the per-point gradient magnitudes (samples off the
// surface carry dY 1e-3 .. 1e-6 of the tile's maximum; e4m3's 14 binades of normals under ONE scale per tile flush them)
__device__ __forceinline__ int bf8_scale_byte(unsigned maxbits) {     // |x| / 2^(E-127) < 2^15 <= 57344 (e5m2 max)
    const int e = (int)((maxbits >> 23) & 0xffu) - 14;
    return e < 1 ? 1 : e;
}
__device__ __forceinline__ void slab_to_bf8(const bf16x8& s, float scale, unsigned& d0, unsigned& d1) {
    union { bf16x8 v; bf16x2v p[4]; } u;
    u.v = s;
    union { s16x2 h; unsigned w; } a, b;
    a.w = 0u;
    b.w = 0u;
    a.h = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(a.h, u.p[0], scale, false);
    a.h = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(a.h, u.p[1], scale, true);
    b.h = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(b.h, u.p[2], scale, false);
    b.h = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(b.h, u.p[3], scale, true);
    d0 = a.w;
    d1 = b.w;
}
__device__ __forceinline__ void save_pair_bf8(int& pending, uint8_t* tile_ptr, int pair, const bf16x8& s0, const bf16x8& s1, int sb,
                                              int lane) {
    const float scale = __uint_as_float((unsigned)sb << 23);
    u32x4 pk;
    unsigned d0, d1;
    slab_to_bf8(s0, scale, d0, d1);
    pk[0] = d0; pk[1] = d1;
    slab_to_bf8(s1, scale, d0, d1);
    pk[2] = d0; pk[3] = d1;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_ptr + (size_t)pair * kPieceBytes, 0, kPieceBytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(pk, rs, (unsigned)lane * 16u, 0, NERFHIP_STORE_AUX);
    pending += 1;
}
__device__ __forceinline__ int bf8_block_scale(float lane_max) { return bf8_scale_byte(wave_max_u32(__float_as_uint(lane_max))); }
#ifndef NERFHIP_F8_DY_E5M2
#define NERFHIP_F8_DY_E5M2 1
#endif
// scale byte of a block from the lanes' partial maxima (one DPP reduction)
__device__ __forceinline__ int f8_block_scale(float lane_max) { return f8_scale_byte(wave_max_u32(__float_as_uint(lane_max))); }

}  // namespace nerfhip
