// render_rays in ONE launch — C ABI (kernel: mlp_render_kernel.h, instantiated per variant in mlp_render_variant.hip).
#include "common.h"
#include "mlp_layout.h"

namespace nerfhip {
using namespace mlp;
typedef nerfhip_render_args RenderArgs;
constexpr int kRenderRays = 4;

template <int PREC, int SV>
int launch_render_variant(const RenderArgs& a, unsigned groups, hipStream_t stream);
template <> int launch_render_variant<NERFHIP_F32, 0>(const RenderArgs&, unsigned, hipStream_t);
template <> int launch_render_variant<NERFHIP_F32, 1>(const RenderArgs&, unsigned, hipStream_t);
template <> int launch_render_variant<NERFHIP_BF16, 0>(const RenderArgs&, unsigned, hipStream_t);
template <> int launch_render_variant<NERFHIP_BF16, 1>(const RenderArgs&, unsigned, hipStream_t);
template <> int launch_render_variant<NERFHIP_BF16, 2>(const RenderArgs&, unsigned, hipStream_t);
template <> int launch_render_variant<NERFHIP_F32, 3>(const RenderArgs&, unsigned, hipStream_t);
template <> int launch_render_variant<NERFHIP_BF16, 3>(const RenderArgs&, unsigned, hipStream_t);

static int render_tail_floats_host(int S_c, int N_i) {
    const int S_f = S_c + N_i;
    const int S4 = (S_c + 3) & ~3, N4 = (N_i + 3) & ~3, F4 = (S_f + 3) & ~3;
    const int fz = N_i > 0 ? 3 * S4 + N4 + ((S_c + 1 + 3) & ~3) + N4 : 0;
    const int coarse = 2 * S4 + fz;
    return coarse > F4 ? coarse : F4;
}

static bool render_shape_ok(int64_t B, int S_c, int N_i, int dtype) {
    if (dtype != NERFHIP_F32 && dtype != NERFHIP_BF16 && dtype != NERFHIP_BF16_F8) return false;
    const int pts = 32 * (dtype == NERFHIP_F32 ? 4 : 8);          // points per sub-pass (KCfg::NW waves of 32)
    if (B <= 0 || B % kRenderRays != 0 || B / kRenderRays > 0x7fffffff || S_c < 3 || N_i < 0) return false;
    if ((kRenderRays * S_c) % pts != 0 || (N_i > 0 && (kRenderRays * (S_c + N_i)) % pts != 0)) return false;
    if ((int64_t)B * (S_c + N_i) / pts > 0x7fffffff) return false;
    // the rays' compositing / depth scratch lives in the weight ring between the sub-passes
    return (size_t)kRenderRays * render_tail_floats_host(S_c, N_i) * sizeof(float) <= (size_t)kSlots * kChunkBytes;
}

static int render_check(const RenderArgs* a, int dtype, bool train) {
    NERFHIP_CHECK_ARG(a != nullptr);
    if (!render_shape_ok(a->B, a->S_c, a->N_i, dtype)) return NERFHIP_E_UNSUPPORTED;
    const bool fine = a->N_i > 0;
    NERFHIP_CHECK_ARG(a->rays && a->packed_coarse && a->z_coarse && a->raw_coarse && a->opacity_coarse);
    NERFHIP_CHECK_ARG(!fine || (a->packed_fine && a->z_fine && a->raw_fine && a->rgb_fine && a->depth_fine && a->opacity_fine));
    NERFHIP_CHECK_ARG(a->perturb <= 0.0f || a->perturb_rand);
    NERFHIP_CHECK_ARG(a->noise_std == 0.0f || (a->noise_coarse && (!fine || a->noise_fine)));
    NERFHIP_CHECK_ARG(a->row_total == NERFHIP_ROW_TOTAL_EXACT || a->row_total == NERFHIP_ROW_TOTAL_ATEN);
    if (train)
        NERFHIP_CHECK_ARG(a->target && a->rgb_coarse && a->depth_coarse && a->g_raw_coarse && a->save_coarse && a->out3 && a->ticket &&
                          (!fine || (a->g_raw_fine && a->save_fine)));
    uintptr_t lines = (uintptr_t)a->z_coarse | (uintptr_t)a->raw_coarse | (uintptr_t)a->z_fine | (uintptr_t)a->raw_fine;
    if (train) lines |= (uintptr_t)a->g_raw_coarse | (uintptr_t)a->g_raw_fine;
    if ((lines & 127) || (((uintptr_t)a->packed_coarse | (uintptr_t)a->packed_fine) & 15)) return NERFHIP_E_ALIGN;
    return 0;
}

}  // namespace nerfhip

extern "C" int nerfhip_render_supported(int64_t B, int S_c, int N_i, int dtype) {
    return nerfhip::render_shape_ok(B, S_c, N_i, dtype) ? 1 : 0;
}

extern "C" int nerfhip_render_fwd(const nerfhip_render_args* args, int dtype, nerfhip_stream_t stream) {
    using namespace nerfhip;
    if (args && args->B == 0) return 0;
    const int rc = render_check(args, dtype, false);
    if (rc) return rc;
    const unsigned groups = (unsigned)(args->B / kRenderRays);
    if (dtype == NERFHIP_F32) return launch_render_variant<NERFHIP_F32, 0>(*args, groups, (hipStream_t)stream);
    return launch_render_variant<NERFHIP_BF16, 0>(*args, groups, (hipStream_t)stream);
}

extern "C" int nerfhip_render_test_fwd(const nerfhip_render_args* args, int dtype, nerfhip_stream_t stream) {
    using namespace nerfhip;
    if (args && args->B == 0) return 0;
    const int rc = render_check(args, dtype, false);
    if (rc) return rc;
    NERFHIP_CHECK_ARG(args->N_i > 0);          // test_time without a fine pass returns nothing but an opacity: not this kernel's business
    const unsigned groups = (unsigned)(args->B / kRenderRays);
    if (dtype == NERFHIP_F32) return launch_render_variant<NERFHIP_F32, 3>(*args, groups, (hipStream_t)stream);
    return launch_render_variant<NERFHIP_BF16, 3>(*args, groups, (hipStream_t)stream);
}

extern "C" int nerfhip_render_train_fwd(const nerfhip_render_args* args, int dtype, nerfhip_stream_t stream) {
    using namespace nerfhip;
    if (args && args->B == 0) return 0;
    const int rc = render_check(args, dtype, true);
    if (rc) return rc;
    const unsigned groups = (unsigned)(args->B / kRenderRays);
    if (dtype == NERFHIP_F32) return launch_render_variant<NERFHIP_F32, 1>(*args, groups, (hipStream_t)stream);
    if (dtype == NERFHIP_BF16_F8) return launch_render_variant<NERFHIP_BF16, 2>(*args, groups, (hipStream_t)stream);
    return launch_render_variant<NERFHIP_BF16, 1>(*args, groups, (hipStream_t)stream);
}
