// The Adam update of torch.optim.Adam (non-amsgrad, L2 weight decay folded into the gradient; reference
// utils/__init__.py:18-20), shared by adam_kernel (optim.hip) and the reduce kernel that applies it in place
// (mlp_bwd.hip) so that both produce the same bits:
//     g  = grad + wd * p;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g g
//     p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#pragma once
#include "common.h"

namespace nerfhip {

struct AdamCoef {
    float step_size, rs2, omb1, omb2;
};
__device__ __forceinline__ AdamCoef adam_coef(float t, float lr, float beta1, float beta2) {
    AdamCoef c;
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    c.step_size = lr / bc1;
    c.rs2 = 1.0f / sqrtf(bc2);
    c.omb1 = 1.0f - beta1;
    c.omb2 = 1.0f - beta2;
    return c;
}
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamCoef& c, float beta2, float eps, float wd) {
    const float gg = g + wd * p;
    m = m + (gg - m) * c.omb1;
    v = v * beta2 + c.omb2 * gg * gg;
    p = p - c.step_size * (m / (sqrtf(v) * c.rs2 + eps));
}

}  // namespace nerfhip
