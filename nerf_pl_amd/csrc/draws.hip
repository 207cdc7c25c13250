// The random draws of one training step in ONE launch, bit for bit what torch's own generators produce.
//
// The reference draws four tensors inside every render_rays call — rand(B,S) jitter (rendering.py:203), randn(B,S) noise (:152),
// rand(B,N_i) importance uniforms (:39), randn(B,S+N_i) noise (:152) — and its DataLoader picks the batch's pixels
// (train.py:89-94, blender.py:81-84).  On the GPU each of these is a ~5 us launch (plus the fill of the generator's graph-safe
// offset when captured): six of the training step's graph nodes.  This kernel produces all of them from the SAME Philox
// stream torch.rand / torch.randn / torch.randint would consume — same seed, same offsets, same values — so a step that uses
// it sees exactly the draws of a step that calls torch, and the caller advances torch's generator by the returned increment.
//
// What is replicated (ATen native/cuda/DistributionTemplates.h on ROCm: hiprand == rocrand's Philox4x32-10):
//   * launch shape: 256-thread blocks, grid = min(max_blocks, ceil(numel / 256)), max_blocks = CUs * (max threads per CU / 256);
//   * thread idx owns Philox subsequence idx at offset `offset` (counter = [offset/4 lo, hi, idx lo, hi], key = seed) and, per
//     round of its grid-stride loop, ONE 4-word block whose word ii belongs to element idx + ii * 256 * grid;
//   * the generator advances by ((numel - 1) / (256 * grid * 4) + 1) * 4 per draw;
//   * uniform:  u = 2^-32 + x * 2^-32  in (0, 1], the value 1 mapped to 0 (ATen's bound reversal);
//   * normal:   Box-Muller on word pairs, (x,y) -> r sin, r cos with r = sqrt(-2 ln(2^-32 + x 2^-32)), angle 2 pi (2^-32 + y 2^-32)
//               through the hardware log2 / sin / cos, exactly as the build of torch in this image evaluates it (philox_normal2);
//   * randint:  (x mod range) as int64; from range 2^28 on ATen joins two words into one 64-bit value (two values per block,
//               unroll factor 2) to bound the modulo bias.
// A randint draw can carry a ray batch: the drawn pixel ids then go straight into ray generation and the colour gather
// (rays.hip: nerfhip_sample_batch) without a round trip through HBM.
//
// Generator state.  Eager calls pass (seed, offset) by value.  A hipGraph replay must see a NEW offset every time, so a
// captured call reads (seed, offset) from a 4 x u64 device buffer `state` = {seed, offset, arrival ticket, -} and the last
// workgroup to finish advances the offset by the call's total increment (every workgroup read the old one first).
#include "draws_body.h"

namespace nerfhip {

__global__ __launch_bounds__(256) void philox_draws_kernel(DrawTable T, unsigned long long seed_v, unsigned long long offset_v,
                                                           unsigned long long* __restrict__ state) {
    philox_draws_block(T, seed_v, offset_v, state, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace nerfhip

extern "C" uint64_t nerfhip_torch_draw_increment(int64_t numel, int max_blocks) {
    if (max_blocks < 1) return 0;
    return (uint64_t)nerfhip::draw_increment(numel, max_blocks);       // (rand / randn / randint below 2^28)
}

extern "C" int nerfhip_torch_draws(const nerfhip_draw* draws_host, int n_draws, const nerfhip_ray_batch* batch_host, uint64_t seed,
                                   uint64_t offset, uint64_t* state, int max_blocks, uint64_t* increment_host,
                                   nerfhip_stream_t stream) {
    nerfhip::DrawTable T;
    int blocks = 0;
    uint64_t inc = 0;
    const int rc = nerfhip_build_draw_table(draws_host, n_draws, batch_host, offset, state, max_blocks, &T, &blocks, &inc);
    if (rc) return rc;
    if (increment_host) *increment_host = inc;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(nerfhip::philox_draws_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, T,
                       (unsigned long long)seed, (unsigned long long)offset, reinterpret_cast<unsigned long long*>(state));
    return nerfhip_launch_status();
}
