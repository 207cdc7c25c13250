// N2: Adam on flat parameter storage — the reference's `Adam(lr, eps=1e-8, weight_decay)` over every
// parameter of both models (utils/__init__.py:10-30 -> torch.optim.Adam, non-amsgrad, L2 weight decay folded into the
// gradient), as ONE launch over up to 8 flat tensors instead of torch's multi-tensor apply (47 us for 1.19 M floats:
// launch/bookkeeping latency, not bandwidth — the update moves 4 x 4.8 MB).
//
//     t      = step + 1                                   (device-resident counter: hipGraph replays advance it)
//     g      = grad + weight_decay * param
//     m      = m + (g - m) (1 - beta1)                    (torch: exp_avg.lerp_(grad, 1 - beta1))
//     v      = beta2 v + (1 - beta2) g g
//     param -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
//
// HBM-bound: 16 B read + 12 B written per element, float4 per thread.  The step counter is advanced by the LAST
// workgroup to finish (arrival ticket), i.e. after every workgroup has read the old value.
#include "adam_math.h"

namespace nerfhip {

constexpr int kAdamMaxTensors = 8;
struct AdamTable {
    float* param[kAdamMaxTensors];
    const float* grad[kAdamMaxTensors];
    float* m[kAdamMaxTensors];
    float* v[kAdamMaxTensors];
    int64_t n[kAdamMaxTensors];
    int block0[kAdamMaxTensors + 1];   // first workgroup of each tensor
    int count;
};

constexpr int kAdamThreads = 256, kAdamVec = 4, kAdamPerBlock = kAdamThreads * kAdamVec * 4;   // 4096 floats / workgroup

__global__ __launch_bounds__(kAdamThreads) void adam_kernel(AdamTable T, float* __restrict__ state, float lr, float beta1,
                                                            float beta2, float eps, float wd) {
    // state[0] = step count (float, exact up to 2^24 steps), state[1] (as unsigned) = arrival ticket
    const float t = state[0] + 1.0f;
    int ti = 0;
#pragma unroll
    for (int k = 1; k < kAdamMaxTensors; ++k) ti += (k < T.count && (int)blockIdx.x >= T.block0[k]) ? 1 : 0;
    const int64_t base = (int64_t)((int)blockIdx.x - T.block0[ti]) * kAdamPerBlock;
    const int64_t n = T.n[ti];
    float* __restrict__ P = T.param[ti];
    const float* __restrict__ G = T.grad[ti];
    float* __restrict__ M = T.m[ti];
    float* __restrict__ V = T.v[ti];
    const AdamCoef ac = adam_coef(t, lr, beta1, beta2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = base + ((int64_t)r * kAdamThreads + threadIdx.x) * kAdamVec;
        if (i + kAdamVec <= n && ((((uintptr_t)(P + i)) | ((uintptr_t)(G + i)) | ((uintptr_t)(M + i)) | ((uintptr_t)(V + i))) & 15) == 0) {
            float4 p = *reinterpret_cast<const float4*>(P + i), g = *reinterpret_cast<const float4*>(G + i);
            float4 m = *reinterpret_cast<const float4*>(M + i), v = *reinterpret_cast<const float4*>(V + i);
            float* pp = &p.x; float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) adam_elem(pp[k], gp[k], mp[k], vp[k], ac, beta2, eps, wd);
            *reinterpret_cast<float4*>(P + i) = p;
            *reinterpret_cast<float4*>(M + i) = m;
            *reinterpret_cast<float4*>(V + i) = v;
        } else {
            for (int k = 0; k < kAdamVec; ++k) {
                const int64_t j = i + k;
                if (j < n) adam_elem(P[j], G[j], M[j], V[j], ac, beta2, eps, wd);
            }
        }
    }
    // arrival ticket: the last workgroup advances the step counter (every workgroup read the old one above)
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* ticket = reinterpret_cast<unsigned*>(state + 1);
        const unsigned prev = atomicAdd(ticket, 1u);
        if (prev == gridDim.x - 1) {
            *ticket = 0u;
            state[0] = t;
        }
    }
}

}  // namespace nerfhip

extern "C" int nerfhip_adam_step(float* const* params_host, const float* const* grads_host, float* const* exp_avg_host,
                                 float* const* exp_avg_sq_host, const int64_t* numel_host, int n_tensors, float* state,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(params_host && grads_host && exp_avg_host && exp_avg_sq_host && numel_host && state);
    NERFHIP_CHECK_ARG(n_tensors >= 1 && n_tensors <= nerfhip::kAdamMaxTensors);
    nerfhip::AdamTable T;
    int blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        NERFHIP_CHECK_ARG(params_host[i] && grads_host[i] && exp_avg_host[i] && exp_avg_sq_host[i] && numel_host[i] > 0);
        T.param[i] = params_host[i];
        T.grad[i] = grads_host[i];
        T.m[i] = exp_avg_host[i];
        T.v[i] = exp_avg_sq_host[i];
        T.n[i] = numel_host[i];
        T.block0[i] = blocks;
        const int64_t nb = (numel_host[i] + nerfhip::kAdamPerBlock - 1) / nerfhip::kAdamPerBlock;
        if (nb + blocks > 0x3fffffff) return NERFHIP_E_BADARG;
        blocks += (int)nb;
    }
    for (int i = n_tensors; i < nerfhip::kAdamMaxTensors; ++i) {
        T.param[i] = nullptr; T.grad[i] = nullptr; T.m[i] = nullptr; T.v[i] = nullptr; T.n[i] = 0;
        T.block0[i] = blocks;
    }
    T.block0[nerfhip::kAdamMaxTensors] = blocks;
    T.count = n_tensors;
    hipLaunchKernelGGL(nerfhip::adam_kernel, dim3(blocks), dim3(nerfhip::kAdamThreads), 0, (hipStream_t)stream, T, state, lr,
                       beta1, beta2, eps, weight_decay);
    return nerfhip_launch_status();
}
