// render_rays in ONE launch (SURVEY §8b "`…_render_fwd(...)` fused: 32 B in + 40 B out per ray"): reference
// models/rendering.py:175-244 — coarse depths, coarse MLP, compositing, sample_pdf + sort, fine MLP, compositing — and, in the
// activation-saving variants, the training step's forward with its loss (train.py:103-117, losses.py:9-14): loss gradient,
// compositing backward and loss / PSNR values.
//
// Mapping.  A workgroup owns a GROUP of 4 whole rays for the entire pipeline.  Its MLP work runs as sub-passes of the fused
// forward's own body (mlp_fwd_kernel.h: one wave = 32 points for the whole network, weights streamed through the LDS ring):
// 4 S_c / P coarse sub-passes, then 4 (S_c + N_i) / P fine ones (P = 256 points for bf16, 128 for fp32: 1 + 3 sub-passes at 64 +
// 128 samples in bf16) — ONE copy of the network code, run in a loop with wave-uniform arguments (weight stream, depths, output
// and activation blocks of the sub-pass's model).  Between the sub-passes the rays' per-point results are ray-complete inside the
// workgroup, so its waves composite them in place (composite_wave.h, one ray per wave; raw comes back from L2, where this
// workgroup's own stores have just put it) and assemble the fine depths (sampling_wave.h) — the same device code, hence the same
// bits, as the stand-alone launches.  The grid is B / 4 workgroups: 256 for the 1024-ray training batch, one per CU.
// Tile numbering of the saved activations is that of the stand-alone forward launches (a group's points are consecutive in
// the flat (B, S) order), so the backward kernels do not know which forward produced their operands.
// SV (the variant): 0 inference | 1 training forward, activations saved in the compute precision | 2 the same with block-scaled e4m3
// copies | 3 inference under test_time (rendering.py:209-213, eval.py:69-79): the COARSE sub-passes run the network's sigma-only
// body (nerf.py:112-114: it stops at the density head), raw_coarse is (B, S_c), only opacity_coarse leaves the coarse pass.
#pragma once
#include "composite_wave.h"
#include "mlp_fwd_kernel.h"

namespace nerfhip {

constexpr int kRenderRays = 4;            // rays per workgroup

typedef nerfhip_render_args RenderArgs;      // include/nerfhip.h

// floats of LDS one ray's tail jobs use: [T_s | w_s | fine_z scratch]
__host__ __device__ inline int render_tail_floats(int S_c, int N_i, int S_f) {
    const int S4 = (S_c + 3) & ~3, N4 = (N_i + 3) & ~3, F4 = (S_f + 3) & ~3;
    const int fz = N_i > 0 ? 3 * S4 + N4 + ((S_c + 1 + 3) & ~3) + N4 : 0;          // fine_z_lds_floats(S_c, N_i)
    const int coarse = 2 * S4 + fz;
    return coarse > F4 ? coarse : F4;
}

// every wave's stores have reached L2 and every wave is here: what one wave wrote, any wave of the workgroup may now read
// (same CU, one L1), and the LDS ring / bias image is free for the next user
__device__ __forceinline__ void render_wg_sync() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The kernel arguments are read where they are used, through the kernarg segment pointer made opaque to the optimiser: left to
// itself hipcc loads all ~45 of them up front and keeps them in SGPRs across the network body (measured: 119-174 SGPRs spilled into
// VGPR lanes, which pushed the body over its 256-register budget: 5-9 VGPRs to scratch).  A fresh pointer per use site keeps each
// scalar load next to its use.
typedef const RenderArgs __attribute__((address_space(4)))* RenderArgsP;
__device__ __forceinline__ RenderArgsP render_args() {
    RenderArgsP p = (RenderArgsP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// compositing of the group's rays in one pass (+ the fine depths, after the coarse pass): R or 2 R one-wave jobs shared out over
// the workgroup's waves; `tail_lds`: the weight ring, idle between sub-passes
template <bool TRAIN, int NW, bool TT = false>
__device__ __forceinline__ void render_tail(const bool fine_pass, const bool with_fine_z, float* const tail_lds, const int wave,
                                            const int lane) {
    constexpr int R = kRenderRays;
    const RenderArgsP a = render_args();
    const int S_c = a->S_c, N_i = a->N_i, S_f = S_c + N_i;
    const float noise_std = a->noise_std;
    const float* raw = fine_pass ? a->raw_fine : a->raw_coarse;
    const float* z = fine_pass ? a->z_fine : a->z_coarse;
    const float* noise = noise_std != 0.0f ? (fine_pass ? a->noise_fine : a->noise_coarse) : nullptr;
    const int S = fine_pass ? S_f : S_c;
    float* rgb = fine_pass ? a->rgb_fine : a->rgb_coarse;
    float* depth = fine_pass ? a->depth_fine : a->depth_coarse;
    float* opac = fine_pass ? a->opacity_fine : a->opacity_coarse;
    const float* rays = a->rays;
    const int white_back = a->white_back;
    const int per_ray = render_tail_floats(S_c, N_i, S_f);
    const int S4 = (S_c + 3) & ~3;
    const int64_t ray0 = (int64_t)blockIdx.x * R, B = a->B;
    const int njobs = with_fine_z ? 2 * R : R;
    for (int job = wave; job < njobs; job += NW) {
        const int slot = job < R ? job : job - R;
        const int64_t r = ray0 + slot;
        if (r >= B) continue;
        float* base = tail_lds + (size_t)slot * per_ray;
        if (job < R) {
            if constexpr (TRAIN) {
                composite_train_wave<true>(raw, z, rays, noise, noise_std, white_back, a->target, a->grad_scale, nullptr, rgb, depth, opac,
                                           fine_pass ? a->g_raw_fine : a->g_raw_coarse, r, S, base, nullptr, lane);
            } else if (TT && !fine_pass) {       // test_time: the coarse pass keeps its opacity only   rendering.py:209-213
                composite_fwd_wave<1>(raw, z, rays, noise, noise_std, white_back, nullptr, nullptr, nullptr, opac, r, S, nullptr, lane);
            } else {
                composite_fwd_wave<4>(raw, z, rays, noise, noise_std, white_back, nullptr, rgb, depth, opac, r, S, nullptr, lane);
            }
        } else {
            // the same ray's weights again (forward sweep only: raw is in L2) and the fine depths from them   rendering.py:223-229
            float* w_s = base + S4;
            composite_weights_wave<(TT ? 1 : 4)>(raw, z, rays, noise, noise_std, r, S, w_s, lane);
            __builtin_amdgcn_wave_barrier();
            const float* u = a->u;
            fine_z_wave(w_s + S4, z + r * S, [&](int j) { return w_s[1 + j]; }, u ? u + r * a->u_stride : nullptr, S, N_i, a->eps,
                        a->z_fine + r * S_f, nullptr, nullptr, nullptr, lane, a->row_total);
        }
    }
}

template <int PREC, int SV>
__global__ __launch_bounds__((KCfg<PREC, (SV == 1 || SV == 2)>::NW * 64), (KCfg<PREC, (SV == 1 || SV == 2)>::WPS))
void mlp_render_kernel(const RenderArgs args_by_value) {          // (read through render_args(), never by name)
    constexpr bool TRAIN = SV == 1 || SV == 2, TT = SV == 3;
    constexpr int FSV = TT ? 0 : SV;                               // the network body's own variant
    constexpr int NW = KCfg<PREC, TRAIN>::NW, PTS = 32 * NW, R = kRenderRays;
    __shared__ __attribute__((aligned(1024))) char lds_all[FwdLds<PREC, TRAIN>::kBytes];
    __shared__ float red[2][16];
    __shared__ unsigned last_s;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* const tail_lds = reinterpret_cast<float*>(lds_all + FwdLds<PREC, TRAIN>::kRingOff);
    int nc, nf;
    {
        const RenderArgsP a = render_args();
        const int S_c = a->S_c, N_i = a->N_i;
        nc = R * S_c / PTS;
        nf = N_i > 0 ? R * (S_c + N_i) / PTS : 0;
    }
    const int total = nc + nf;
#pragma clang loop unroll(disable)
    for (int sp = 0; sp < total; ++sp) {
        if (sp > 0) render_wg_sync();
        if (sp == nc) {                   // (only with a fine pass) the coarse pass is complete for this group's rays
            render_tail<TRAIN, NW, TT>(false, true, tail_lds, wave, lane);
            render_wg_sync();
        }
        const RenderArgsP a = render_args();
        const bool fine = sp >= nc;
        const int S = fine ? a->S_c + a->N_i : a->S_c;
        const unsigned blk = fine ? blockIdx.x * (unsigned)nf + (unsigned)(sp - nc) : blockIdx.x * (unsigned)nc + (unsigned)sp;
        const FwdZGen zg{fine ? nullptr : a->perturb_rand, fine ? nullptr : a->z_coarse, fine ? 0 : a->use_disp, fine ? 0.0f : a->perturb,
                          TRAIN ? a->regen_enc : 0};
        uint8_t* save = nullptr;
        if constexpr (TRAIN) save = (uint8_t*)(fine ? a->save_fine : a->save_coarse);
        if (TT && !fine)                  // (wave-uniform) the coarse network up to the density head: out = sigma (B, S_c)
            mlp_fwd_body<PREC, MODE_RAYS, true, 0>(lds_all, blk, a->rays, nullptr, a->B * (int64_t)S, (int64_t)S, (const uint8_t*)a->packed_coarse,
                                                   a->raw_coarse, nullptr, zg);
        else
            mlp_fwd_body<PREC, MODE_RAYS, false, FSV>(lds_all, blk, a->rays, fine ? a->z_fine : nullptr, a->B * (int64_t)S, (int64_t)S,
                                                      (const uint8_t*)(fine ? a->packed_fine : a->packed_coarse), fine ? a->raw_fine : a->raw_coarse, save, zg);
    }
    render_wg_sync();
    render_tail<TRAIN, NW, TT>(nf > 0, false, tail_lds, wave, lane);

    if constexpr (TRAIN) {
        // loss / PSNR values (mse_psnr_kernel's own order, loss_math.h): every workgroup announces its rays' colours — device-scope
        // write-through stores — with an arrival ticket; the last one reduces the two images against the target
        const RenderArgsP a = render_args();
        unsigned* const ticket = a->ticket;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_s = (prev == gridDim.x - 1) ? 1u : 0u;
            if (prev == gridDim.x - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!last_s) return;
        // consumer side of the hand-off: an agent-scope ACQUIRE in the one workgroup that reads the others' colours.  The
        // producer side stays write-through stores + s_waitcnt vmcnt(0) + the relaxed ticket: an agent-scope RELEASE there writes this
        // XCD's whole L2 back once per workgroup (measured: 20 us per launch instead of 7, round 4).
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float* img_c = a->rgb_coarse;
        const float* img_f = a->rgb_fine;
        const float* target = a->target;
        float* out3 = a->out3;
        const int64_t n3 = 3 * a->B;
        auto fresh_c = [&](int64_t i) { return __hip_atomic_load(img_c + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto fresh_f = [&](int64_t i) { return __hip_atomic_load(img_f + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        constexpr int Q = 1024 / (NW * 64);
        if (nf > 0)
            mse_psnr_block<Q>(fresh_c, fresh_f, true, target, n3, out3, nullptr, nullptr, red);
        else
            mse_psnr_block<Q>(fresh_c, [&](int64_t i) { return 0.0f; }, false, target, n3, out3, nullptr, nullptr, red);
    }
}

template <int PREC, int SV>
int launch_render_variant(const RenderArgs& a, unsigned groups, hipStream_t stream);

}  // namespace nerfhip
