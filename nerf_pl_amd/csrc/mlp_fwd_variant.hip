// One instantiation of the fused MLP forward kernel (see mlp_fwd_kernel.h); compiled by nerf_pl_amd/build.py once per
//   -DNH_PREC={0 fp32,1 bf16} -DNH_MODE={0 embedded,1 rays} -DNH_VARIANT={0 inference, 1 sigma-only, 2 activation-saving, 3 activation-saving as e4m3 (bf16 only)}
#include "mlp_fwd_kernel.h"

#if !defined(NH_PREC) || !defined(NH_MODE) || !defined(NH_VARIANT)
#error "compile with -DNH_PREC= -DNH_MODE= -DNH_VARIANT= (nerf_pl_amd/build.py)"
#endif

namespace nerfhip {

constexpr int kSV = NH_VARIANT >= 2 ? NH_VARIANT - 1 : 0;    // 0 none, 1 native, 2 e4m3
static_assert(KCfg<NH_PREC, kSV != 0>::NW == (NH_PREC == NERFHIP_BF16 ? 8 : 4), "mlp_fwd.hip: fwd_waves() out of sync");

template <>
int launch_fwd_variant<NH_PREC, NH_MODE, NH_VARIANT == 1, kSV>(const float* in0, const float* in1, int64_t n, int64_t aux,
                                                               const void* packed, float* out, void* save, unsigned blocks,
                                                               hipStream_t stream, const FwdZGen& zg) {
    constexpr int NW = KCfg<NH_PREC, kSV != 0>::NW;
    hipLaunchKernelGGL((mlp_fwd_kernel<NH_PREC, NH_MODE, NH_VARIANT == 1, kSV>), dim3(blocks), dim3(NW * 64), 0, stream,
                       in0, in1, n, aux, (const uint8_t*)packed, out, (uint8_t*)save, zg);
    return nerfhip_launch_status();
}

}  // namespace nerfhip
