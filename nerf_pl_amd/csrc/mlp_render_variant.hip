// One instantiation of the single-launch render kernel (mlp_render_kernel.h); compiled by nerf_pl_amd/build.py once per
//   -DNH_PREC={0 fp32,1 bf16} -DNH_SV={0 inference, 1 training forward (activations saved in the compute precision), 2 the same with
//   block-scaled e4m3 copies (bf16 only), 3 inference under test_time (sigma-only coarse sub-passes)}
#include "mlp_render_kernel.h"

#if !defined(NH_PREC) || !defined(NH_SV)
#error "compile with -DNH_PREC= -DNH_SV= (nerf_pl_amd/build.py)"
#endif

namespace nerfhip {

template <>
int launch_render_variant<NH_PREC, NH_SV>(const RenderArgs& a, unsigned groups, hipStream_t stream) {
    constexpr int NW = KCfg<NH_PREC, (NH_SV == 1 || NH_SV == 2)>::NW;
    hipLaunchKernelGGL((mlp_render_kernel<NH_PREC, NH_SV>), dim3(groups), dim3(NW * 64), 0, stream, a);
    return nerfhip_launch_status();
}

}  // namespace nerfhip
