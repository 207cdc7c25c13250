// fp8 (OCP e4m3, block-scaled) storage of the tensors saved for the weight-gradient GEMM — shared by the activation-saving
// forward (mlp_fwd_kernel.h) and the backward chain (mlp_bwd.hip).  Format: mlp_layout.h "fp8 storage of the saved tensors".
#pragma once
#include "common.h"
#include "mlp_layout.h"

#ifndef NERFHIP_STORE_AUX
#define NERFHIP_STORE_AUX 2     // cache-policy bits of the write-once stores: 2 = nt
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
// Scale table: one DWORD per pair (the e8m0 byte, zero-extended) at dword index f8_x_scale_pos / f8_dy_scale_pos — the dW
// kernel DMAs single dwords of it into LDS and hands them to the MFMA as scale operands (byte 0).  The <= 8 scales of a
// 16-slab section are wave-uniform (SGPRs); lane 0 writes them as two 16-byte stores when the section is complete.
struct F8Scales {
    int v[8];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0;
    }
    __device__ __forceinline__ void set(int idx, int byte) { v[idx] = byte; }      // idx: compile-time constant
};
__device__ __forceinline__ void save_scales_f8(int& pending, uint8_t* tile_ptr, int scale_off, int pos0, const F8Scales& sc,
                                               int lane) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_ptr + scale_off + 4 * pos0, 0, 32, 0x00020000);
    u32x4 lo, hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] = (unsigned)sc.v[i];
        hi[i] = (unsigned)sc.v[4 + i];
    }
    // lanes 1..63 fall outside the 32-byte descriptor and are dropped by the bounds check: no exec juggling needed
    __builtin_amdgcn_raw_buffer_store_b128(lo, rs, (unsigned)lane * 32u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(hi, rs, (unsigned)lane * 32u + 16u, 0, 0);
    pending += 2;
}
// store slab pair `pair` (slabs 2*pair, 2*pair+1; `mx` = this lane's max |value| over both) as one e4m3 piece; returns the
// pair's e8m0 scale byte (wave-uniform)
__device__ __forceinline__ int save_pair_f8(int& pending, uint8_t* tile_ptr, int pair, const bf16x8& s0, const bf16x8& s1,
                                            float mx, int lane) {
    const int sb = f8_scale_byte(wave_max_u32(__float_as_uint(mx)));
    const float scale = __uint_as_float((unsigned)sb << 23);
    u32x4 pk;
    unsigned d0, d1;
    slab_to_f8(s0, scale, d0, d1);
    pk[0] = d0; pk[1] = d1;
    slab_to_f8(s1, scale, d0, d1);
    pk[2] = d0; pk[3] = d1;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_ptr + (size_t)pair * kPieceBytes, 0, kPieceBytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(pk, rs, (unsigned)lane * 16u, 0, NERFHIP_STORE_AUX);     // soffset 0: see save_slabs
    pending += 1;
    return sb;
}

}  // namespace nerfhip
