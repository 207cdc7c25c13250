// Device body + launch-table builder of the torch-compatible draws (see draws.hip for what is replicated and why), shared by
// nerfhip_torch_draws (draws.hip) and the training step's one-launch prologue (prologue.hip).
#pragma once
#include "rays_math.h"

namespace nerfhip {

constexpr int kDrawMax = 6;
constexpr int kDrawBlocks = 64;          // workgroups of one launch per draw
constexpr unsigned long long kRandint64From = 1ull << 28;   // ATen random_from_to: ranges from 2^28 take 64-bit words (two per Philox block)
struct RayBatch {
    const float* c2w;
    const float* rgbs_all;
    float* rays;
    float* rgbs;
    int H, W;
    float focal, near, far;
    int use_ndc;
    float ndc_plane, sx, sy;
};
struct DrawSeg {
    void* out;
    int64_t numel;
    unsigned long long rel;                // Philox offset of the draw relative to the call's offset
    unsigned long long range;
    int block0, nblocks, grid;             // first workgroup / workgroups of this launch serving the draw; ATen's grid for it
    int kind, batch;                       // NERFHIP_DRAW_*; batch != 0: this randint draw feeds the ray batch
};
struct DrawTable {
    DrawSeg seg[kDrawMax];
    int count;
    unsigned long long total;              // the call's total increment
    RayBatch batch;
};

// Philox4x32-10 (Random123 constants), one 4-word block
__device__ __forceinline__ uint4 philox_block(uint4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__device__ __forceinline__ float philox_uniform(unsigned x) {
#pragma clang fp contract(fast)
    const float v = 2.3283064e-10f + ((float)x * 2.3283064e-10f);        // rocrand uniform_distribution: (0, 1]
    const float value = v * 1.0f + 0.0f;                                  // ATen uniform_: rand * (to - from) + from
    return value == 1.0f ? 0.0f : value;                                  // ... and its bound reversal: [0, 1)
}
// Box-Muller as torch.randn runs it (rocrand's box_muller as compiled into torch 2.10 + ROCm 7.0, established bit for bit against
// torch.randn by tools/probes/probe_randn*.py: 0 mismatches in 2 x 40,000 values, both branches):
//   u = 2^-32 + x 2^-32,  angle = fma(y, c, c) with c = 2 pi 2^-32,
//   ln u = hardware log2 (v_log_f32) times ln 2 as a compensated product (hi / lo split of ln 2, two fmas),
//   r = correctly rounded sqrt(-2 ln u),  (r sin, r cos) through the hardware sin / cos.
// (ROCm 7.2's own logf is a different, more accurate algorithm: it disagrees with torch's in 15 % of the values.)
__device__ __forceinline__ void philox_normal2(unsigned x, unsigned y, float& a, float& b) {
#pragma clang fp contract(off)
    const float u = 2.3283064e-10f + ((float)x * 2.3283064e-10f);
    const float v = __builtin_fmaf((float)y, 1.46291807e-09f, 1.46291807e-09f);
    const float l2 = __log2f(u);                                          // u >= 2^-32: never a denormal
    const float ln2_hi = 0x1.62e42ep-1f, ln2_lo = 0x1.efa39ep-25f;
    const float r0 = l2 * ln2_hi;
    const float lnu = r0 + __builtin_fmaf(l2, ln2_lo, __builtin_fmaf(l2, ln2_hi, -r0));
    const float s = sqrtf(-2.0f * lnu);
    float sn, cs;
    __sincosf(v, &sn, &cs);
    a = sn * s;                                                           // (ATen normal_: rand * 1 + 0)
    b = cs * s;
}

// the work of workgroup `bid` of the `nblocks` draw workgroups of a launch (256 threads)
__device__ __forceinline__ void philox_draws_block(const DrawTable& T, unsigned long long seed_v, unsigned long long offset_v,
                                                   unsigned long long* __restrict__ state, int bid, int nblocks) {
    // this workgroup's draw (wave-uniform selects over the kernel arguments: no dynamically indexed copy of the table)
    DrawSeg sg = T.seg[0];
#pragma unroll
    for (int k = 1; k < kDrawMax; ++k)
        if (k < T.count && bid >= T.seg[k].block0) sg = T.seg[k];
    const unsigned long long seed = state ? state[0] : seed_v;
    const unsigned long long base = state ? state[1] : offset_v;
    const unsigned long long off = base + sg.rel;
    const int64_t G = (int64_t)256 * sg.grid;
    const int64_t numel = sg.numel;
    const int kind = sg.kind;
    const bool wide = kind == NERFHIP_DRAW_RANDINT && sg.range >= kRandint64From;      // two 64-bit values per Philox block
    const int U = wide ? 2 : 4;                                                        // ATen's unroll factor
    const int64_t rounded = numel > 0 ? ((numel - 1) / (G * U) + 1) * G * U : 0;
    const unsigned long long c0 = off >> 2;
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    // ATen launches `grid` workgroups for this draw; this launch serves them with at most kDrawBlocks of its own, each
    // standing in for ATen's workgroups vb, vb + nblocks, ... (fewer arrival tickets on one address: 772 of them cost 7 us)
    for (int vb = bid - sg.block0; vb < sg.grid; vb += sg.nblocks) {
    const int64_t idx = (int64_t)vb * 256 + threadIdx.x;
    // rocrand_init(seed, subsequence = idx, offset): counter = [offset / 4, idx]
    uint4 ctr = make_uint4((unsigned)c0, (unsigned)(c0 >> 32), (unsigned)idx, (unsigned)((unsigned long long)idx >> 32));
    for (int64_t li0 = idx; li0 < rounded; li0 += G * U) {
        const uint4 r = philox_block(ctr, k0, k1);
        ctr.x += 1u;                                                       // next4(): bump the 128-bit counter
        if (ctr.x == 0u) { ctr.y += 1u; if (ctr.y == 0u) { ctr.z += 1u; if (ctr.z == 0u) ctr.w += 1u; } }
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        if (kind == NERFHIP_DRAW_UNIFORM) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) f[ii] = philox_uniform(w[ii]);
        } else if (kind == NERFHIP_DRAW_NORMAL) {
            philox_normal2(w[0], w[1], f[0], f[1]);
            philox_normal2(w[2], w[3], f[2], f[3]);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int64_t li = li0 + G * ii;
            if (ii >= U || li >= numel) continue;
            if (kind == NERFHIP_DRAW_RANDINT) {
                // ATen uniform_int_from_to: (word mod range) + 0; the wide path joins two words, high word first
                const unsigned long long word = wide ? (((unsigned long long)w[2 * (ii & 1)] << 32) | w[2 * (ii & 1) + 1]) : (unsigned long long)w[ii];
                const int64_t id = (int64_t)(word % sg.range);
                if (sg.out) reinterpret_cast<int64_t*>(sg.out)[li] = id;
                if (sg.batch) {
                    const RayBatch& b = T.batch;
                    if (b.rgbs) {
                        b.rgbs[3 * li] = b.rgbs_all[3 * id];
                        b.rgbs[3 * li + 1] = b.rgbs_all[3 * id + 1];
                        b.rgbs[3 * li + 2] = b.rgbs_all[3 * id + 2];
                    }
                    gen_ray(b.c2w, id, b.H, b.W, b.focal, b.near, b.far, b.use_ndc, b.ndc_plane, b.sx, b.sy, b.rays + li * 8);
                }
            } else {
                reinterpret_cast<float*>(sg.out)[li] = f[ii];
            }
        }
    }
    }
    if (state) {       // arrival ticket: the last workgroup advances the offset (every workgroup read the old one above)
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long prev = atomicAdd(state + 2, 1ull);
            if (prev == (unsigned long long)nblocks - 1ull) {
                state[2] = 0ull;
                state[1] = base + T.total;
            }
        }
    }
}

static inline int draw_grid(int64_t numel, int max_blocks) {
    const int64_t need = (numel + 255) / 256;
    return (int)(need < (int64_t)max_blocks ? need : (int64_t)max_blocks);
}
static inline unsigned long long draw_increment(int64_t numel, int max_blocks, int unroll = 4) {
    if (numel <= 0) return 0ull;                                           // ATen returns before touching the generator
    const int64_t per_round = (int64_t)256 * draw_grid(numel, max_blocks) * unroll;
    return (unsigned long long)(((numel - 1) / per_round + 1) * 4);
}
static inline int draw_unroll(const nerfhip_draw& d) {
    return (d.kind == NERFHIP_DRAW_RANDINT && d.range >= kRandint64From) ? 2 : 4;
}

}  // namespace nerfhip

// The launch table of nerfhip_torch_draws' arguments (shared with the training-step prologue launch, prologue.hip).
// Returns 0 and sets *blocks (0: nothing to launch) and *increment, or a NERFHIP_E_* code.
static inline int nerfhip_build_draw_table(const nerfhip_draw* draws_host, int n_draws, const nerfhip_ray_batch* batch_host,
                                           uint64_t offset, const uint64_t* state, int max_blocks, nerfhip::DrawTable* out,
                                           int* blocks_out, uint64_t* increment_out) {
    NERFHIP_CHECK_ARG(draws_host && n_draws >= 1 && n_draws <= nerfhip::kDrawMax && max_blocks >= 1);
    NERFHIP_CHECK_ARG(state || (offset & 3) == 0);
    nerfhip::DrawTable T{};
    T.count = 0;
    int blocks = 0;
    unsigned long long rel = 0;
    for (int i = 0; i < n_draws; ++i) {
        const nerfhip_draw& d = draws_host[i];
        NERFHIP_CHECK_ARG(d.numel >= 0 && d.kind >= NERFHIP_DRAW_UNIFORM && d.kind <= NERFHIP_DRAW_RANDINT);
        const bool with_batch = (batch_host && i == 0);
        if (with_batch) NERFHIP_CHECK_ARG(d.kind == NERFHIP_DRAW_RANDINT);
        if (d.numel == 0) continue;
        if (d.kind == NERFHIP_DRAW_RANDINT) NERFHIP_CHECK_ARG(d.range >= 1 && d.range <= (1ull << 62));
        const unsigned long long inc = nerfhip::draw_increment(d.numel, max_blocks, nerfhip::draw_unroll(d));
        if (!d.out && !with_batch) {           // a draw nobody reads: the stream moves past it, nothing is launched for it
            rel += inc;
            continue;
        }
        nerfhip::DrawSeg& g = T.seg[T.count++];
        g.out = d.out;
        g.numel = d.numel;
        g.rel = rel;
        g.range = d.kind == NERFHIP_DRAW_RANDINT ? d.range : 1ull;
        g.kind = d.kind;
        g.block0 = blocks;
        g.grid = nerfhip::draw_grid(d.numel, max_blocks);
        g.nblocks = g.grid < nerfhip::kDrawBlocks ? g.grid : nerfhip::kDrawBlocks;
        g.batch = with_batch ? 1 : 0;
        blocks += g.nblocks;
        rel += inc;
        if (with_batch) {
            const nerfhip_ray_batch& b = *batch_host;
            NERFHIP_CHECK_ARG(b.c2w && b.rays && b.H > 0 && b.W > 0 && ((b.rgbs == nullptr) == (b.rgbs_all == nullptr)));
            if (((uintptr_t)b.rays) & 15) return NERFHIP_E_ALIGN;
            T.batch.c2w = b.c2w; T.batch.rgbs_all = b.rgbs_all; T.batch.rays = b.rays; T.batch.rgbs = b.rgbs;
            T.batch.H = b.H; T.batch.W = b.W; T.batch.focal = (float)b.focal; T.batch.near = b.near; T.batch.far = b.far;
            T.batch.use_ndc = b.use_ndc; T.batch.ndc_plane = b.ndc_near_plane;
            T.batch.sx = nerfhip_ndc_scale(b.W, b.focal); T.batch.sy = nerfhip_ndc_scale(b.H, b.focal);
        }
    }
    T.total = rel;
    *increment_out = (uint64_t)rel;
    *blocks_out = 0;
    *out = T;
    if (T.count == 0 && !(state && rel)) return 0;
    if (T.count == 0) {                        // only skipped draws, captured: one workgroup moves the device offset on
        T.count = 1;
        T.seg[0].grid = 1;
        T.seg[0].nblocks = 1;
        blocks = 1;
    }
    *out = T;
    *blocks_out = blocks;
    return 0;
}
