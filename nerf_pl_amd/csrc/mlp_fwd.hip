// K2: fused NeRF MLP forward — C ABI (kernels: mlp_fwd_kernel.h, instantiated per variant in mlp_fwd_variant.hip).
#include "common.h"
#include "mlp_layout.h"

namespace nerfhip {
using namespace mlp;
constexpr int MODE_EMBEDDED = 0, MODE_RAYS = 1;

template <int PREC, int MODE, bool SIGMA_ONLY, int SV>      // SV: 0 inference, 1 save (compute precision), 2 save as e4m3
int launch_fwd_variant(const float* in0, const float* in1, int64_t n, int64_t aux, const void* packed, float* out, void* save,
                       unsigned blocks, hipStream_t stream, const FwdZGen& zg);
#define NH_ARGS const float*, const float*, int64_t, int64_t, const void*, float*, void*, unsigned, hipStream_t, const FwdZGen&
#define NH_DECL(P, M)                                                  \
    template <> int launch_fwd_variant<P, M, false, 0>(NH_ARGS);   \
    template <> int launch_fwd_variant<P, M, true, 0>(NH_ARGS);    \
    template <> int launch_fwd_variant<P, M, false, 1>(NH_ARGS);
NH_DECL(NERFHIP_F32, 0) NH_DECL(NERFHIP_F32, 1) NH_DECL(NERFHIP_BF16, 0) NH_DECL(NERFHIP_BF16, 1)
template <> int launch_fwd_variant<NERFHIP_BF16, 0, false, 2>(NH_ARGS);
template <> int launch_fwd_variant<NERFHIP_BF16, 1, false, 2>(NH_ARGS);
#undef NH_ARGS
#undef NH_DECL

// waves per workgroup of each variant (must match KCfg in mlp_fwd_kernel.h; checked there by static_assert)
constexpr int fwd_waves(int prec, bool save) { return prec == NERFHIP_BF16 ? 8 : 4; }

template <int PREC, int MODE>
static int launch_fwd(const float* in0, const float* in1, int64_t n, int64_t aux, const void* packed, float* out,
                      int sigma_only, void* save, bool save_f8, hipStream_t stream, const FwdZGen& zg = FwdZGen{nullptr, nullptr, 0, 0.0f}) {
    constexpr int NW = fwd_waves(PREC, false);
    const int64_t blocks = (n + 32 * NW - 1) / (32 * NW);
    if (blocks > 0x7fffffff) return NERFHIP_E_BADARG;
    if (save) {
        if (sigma_only) return NERFHIP_E_UNSUPPORTED;
        // every tile of the (workgroup-padded) activation block must be written: the backward reads them all
        constexpr int NWS = fwd_waves(PREC, true);
        const int64_t ppw = 32 * (PREC == NERFHIP_BF16 ? 8 : 4);                  // padding unit of nerfhip_mlp_act_bytes
        const int64_t tiles = (n + ppw - 1) / ppw * (ppw / 32);
        if constexpr (PREC == NERFHIP_BF16) {
            if (save_f8)
                return launch_fwd_variant<PREC, MODE, false, 2>(in0, in1, n, aux, packed, out, save, (unsigned)(tiles / NWS), stream, zg);
        }
        return launch_fwd_variant<PREC, MODE, false, 1>(in0, in1, n, aux, packed, out, save, (unsigned)(tiles / NWS), stream, zg);
    }
    if (sigma_only) return launch_fwd_variant<PREC, MODE, true, 0>(in0, in1, n, aux, packed, out, nullptr, (unsigned)blocks, stream, zg);
    return launch_fwd_variant<PREC, MODE, false, 0>(in0, in1, n, aux, packed, out, nullptr, (unsigned)blocks, stream, zg);
}

}  // namespace nerfhip

extern "C" size_t nerfhip_mlp_act_bytes(int64_t n_points, int dtype) {
    if (n_points < 0 || (dtype != NERFHIP_F32 && dtype != NERFHIP_BF16 && dtype != NERFHIP_BF16_F8)) return 0;
    const int64_t ppw = 32 * (dtype == NERFHIP_F32 ? 4 : 8);                  // points per workgroup
    const int64_t tiles = (n_points + ppw - 1) / ppw * (ppw / 32);             // whole workgroups are written
    return (size_t)tiles * (dtype == NERFHIP_BF16_F8 ? nerfhip::mlp::f8_act_tile_bytes() : nerfhip::mlp::act_tile_bytes(dtype));
}

extern "C" int nerfhip_mlp_fwd_embedded(const float* x, int64_t x_stride, int64_t n, const void* packed, float* out,
                                        int sigma_only, int dtype, void* save_acts, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0 && x_stride >= (sigma_only ? 63 : 90));
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(x && packed && out);
    if ((((uintptr_t)packed) & 15) || (!sigma_only && (((uintptr_t)out) & 15))) return NERFHIP_E_ALIGN;
    if (dtype == NERFHIP_BF16 || dtype == NERFHIP_BF16_F8)
        return nerfhip::launch_fwd<NERFHIP_BF16, nerfhip::MODE_EMBEDDED>(x, nullptr, n, x_stride, packed, out, sigma_only,
                                                                          save_acts, dtype == NERFHIP_BF16_F8, (hipStream_t)stream);
    if (dtype == NERFHIP_F32)
        return nerfhip::launch_fwd<NERFHIP_F32, nerfhip::MODE_EMBEDDED>(x, nullptr, n, x_stride, packed, out, sigma_only,
                                                                         save_acts, false, (hipStream_t)stream);
    return NERFHIP_E_UNSUPPORTED;
}

extern "C" int nerfhip_mlp_fwd_rays(const float* rays, const float* z, int64_t B, int S, const void* packed, float* out,
                                    int sigma_only, int dtype, void* save_acts, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S >= 1);
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(rays && z && packed && out);
    if ((((uintptr_t)packed) & 15) || (!sigma_only && (((uintptr_t)out) & 15))) return NERFHIP_E_ALIGN;
    const int64_t n = B * (int64_t)S;
    if (dtype == NERFHIP_BF16 || dtype == NERFHIP_BF16_F8)
        return nerfhip::launch_fwd<NERFHIP_BF16, nerfhip::MODE_RAYS>(rays, z, n, S, packed, out, sigma_only, save_acts,
                                                                      dtype == NERFHIP_BF16_F8, (hipStream_t)stream);
    if (dtype == NERFHIP_F32)
        return nerfhip::launch_fwd<NERFHIP_F32, nerfhip::MODE_RAYS>(rays, z, n, S, packed, out, sigma_only, save_acts, false,
                                                                     (hipStream_t)stream);
    return NERFHIP_E_UNSUPPORTED;
}

extern "C" int nerfhip_mlp_fwd_rays_coarse(const float* rays, const float* perturb_rand, float* z, int64_t B, int S, int use_disp,
                                           float perturb, const void* packed, float* out, int sigma_only, int dtype, void* save_acts,
                                           nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(B >= 0 && S >= 1);
    if (B == 0) return 0;
    NERFHIP_CHECK_ARG(rays && z && packed && out && (perturb <= 0.0f || perturb_rand));
    if ((((uintptr_t)packed) & 15) || (!sigma_only && (((uintptr_t)out) & 15))) return NERFHIP_E_ALIGN;
    const int64_t n = B * (int64_t)S;
    const nerfhip::FwdZGen zg{perturb > 0.0f ? perturb_rand : nullptr, z, use_disp, perturb};
    if (dtype == NERFHIP_BF16 || dtype == NERFHIP_BF16_F8)
        return nerfhip::launch_fwd<NERFHIP_BF16, nerfhip::MODE_RAYS>(rays, nullptr, n, S, packed, out, sigma_only, save_acts,
                                                                      dtype == NERFHIP_BF16_F8, (hipStream_t)stream, zg);
    if (dtype == NERFHIP_F32)
        return nerfhip::launch_fwd<NERFHIP_F32, nerfhip::MODE_RAYS>(rays, nullptr, n, S, packed, out, sigma_only, save_acts, false,
                                                                     (hipStream_t)stream, zg);
    return NERFHIP_E_UNSUPPORTED;
}
