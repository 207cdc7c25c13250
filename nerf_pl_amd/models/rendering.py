"""`models.rendering` of kwea123/nerf_pl, MI355X-native.

`render_rays` keeps the reference signature, argument meaning, returned keys and RNG consumption
order (models/rendering.py:58-244) but runs as 5 HIP launches per call:
    sample_coarse_z -> mlp(coarse) -> composite -> fine_z (sample_pdf + merge) -> mlp(fine) -> composite
with points, encodings and the repeated direction embedding never materialised in HBM.

That is the reference's configuration (NeRF D=8 W=256 skips=[4], Embedding(3,10)/(3,4) logscale: train.py:34-42).  Any other
NeRF shape or embedding keeps the same pipeline with the MLP stage unfused (`_mlp_points`): points -> Embedding (posenc
kernel) -> NeRF.forward layer by layer (csrc/linear.hip), in point chunks of `chunk` like the reference's loop
(rendering.py:115-141).
"""
import os

import torch

from .. import draws as D
from .. import ops
from .mlp_autograd import _needs_grad, mlp_rays

# test_time renders (eval.py: coarse pass sigma-only, rendering.py:209-213) through the single-launch kernel too?  Since round 6 its
# coarse sub-passes run the network's sigma-only body (nerfhip_render_test_fwd): exactly the FLOPs of the five launches, bit-identical
# results (tests/test_gpu_render_fused.py).  Measured on MI355X, 800 x 800 image in 32768-ray chunks incl. D2H, same box, alternating
# (profiles/r06_eval_single_launch_ab.txt): 136.9 / 136.9 ms single launch against 135.8 / 135.8 ms for the launches — at 8192
# workgroups per chunk the launches each fill the chip, while a workgroup of the single launch idles its MFMA pipe during its
# compositing phases (one workgroup per CU: nothing else to run there) — so the launches stay the default for test_time;
# NERFHIP_FUSE_TEST_TIME=1 selects the single launch (one graph node per chunk instead of five).
FUSE_TEST_TIME = os.environ.get("NERFHIP_FUSE_TEST_TIME", "0") == "1"

__all__ = ['render_rays']


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5):
    """Sample N_importance depths from the piecewise-constant pdf `weights` over `bins`.
    Reference: models/rendering.py:14-55 (cumsum + torchsearchsorted + gather + lerp) -> one launch.

    bins: (N_rays, N_samples_+1), weights: (N_rays, N_samples_) -> (N_rays, N_importance)."""
    N_rays = weights.shape[0]
    u = None
    if not det:
        u = torch.rand(N_rays, N_importance, device=bins.device)   # same draw as rendering.py:39
    return ops.sample_pdf_u(bins.float(), weights.float(), N_importance, u=u, eps=eps)


def _fusable(models, embeddings):
    from .nerf import Embedding, NeRF
    ok = all(isinstance(m, NeRF) and m.is_default_arch() for m in models)
    ok = ok and len(embeddings) >= 2 and all(isinstance(e, Embedding) and e.logscale for e in embeddings[:2])
    return ok and embeddings[0].N_freqs == 10 and embeddings[1].N_freqs == 4 \
        and embeddings[0].in_channels == 3 and embeddings[1].in_channels == 3


def _mlp_points(model, embedding_xyz, rays, z, dir_embedded, sigma_only, chunk):
    """raw (B,S,4) [or sigma (B,S)] of `model` at the points o + d z: the unfused MLP stage (rendering.py:115-141, :206).
    Each chunk of `chunk` points is embedded, joined with the direction embedding of the ray that owns the point (a gather:
    the S-fold repeat of rendering.py:119 is never materialised) and evaluated by NeRF.forward."""
    B, S = z.shape
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]).reshape(-1, 3)
    n = B * S
    outs = []
    for lo in range(0, n, max(1, chunk)):
        hi = min(n, lo + max(1, chunk))
        e = embedding_xyz(pts[lo:hi])
        if not sigma_only:
            owner = torch.arange(lo, hi, device=z.device) // S
            e = torch.cat([e, dir_embedded.index_select(0, owner)], 1)
        outs.append(model(e, sigma_only=sigma_only))
    out = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
    return out.view(B, S) if sigma_only else out.view(B, S, 4)


def render_rays(models,
                embeddings,
                rays,
                N_samples=64,
                use_disp=False,
                perturb=0,
                noise_std=1,
                N_importance=0,
                chunk=1024*32,
                white_back=False,
                test_time=False
                ):
    """Render rays by computing the output of @model applied on @rays.
    Same contract as the reference (models/rendering.py:58-85):

    models: [coarse NeRF(, fine NeRF)], embeddings: [xyz Embedding, dir Embedding],
    rays: (N_rays, 3+3+2) origins, directions, near, far.
    Returns dict with rgb_coarse/depth_coarse (unless test_time), opacity_coarse and, when
    N_importance>0, rgb_fine/depth_fine/opacity_fine.
    `chunk`: the fused kernel needs no point-chunk loop (its only per-point HBM footprint is 4 B in + 16 B out); the
    layer-by-layer path of non-default shapes evaluates `chunk` points per MLP call, as the reference does.
    """
    rays = rays.float().contiguous()
    N_rays = rays.shape[0]
    dev = rays.device
    model_coarse = models[0]
    if _fusable(models, embeddings):
        def mlp(model, z, sigma_only):
            return mlp_rays(model, rays, z, sigma_only=sigma_only)
    else:
        dir_embedded = embeddings[1](rays[:, 3:6])                                          # :186 (raw rays_d)

        def mlp(model, z, sigma_only):
            return _mlp_points(model, embeddings[0], rays, z, dir_embedded, sigma_only, int(chunk))

    # ---- the whole call in ONE launch (nerfhip_render_fwd) where nothing needs a gradient and the shape fits the kernel's ray
    # groups: the same four draws in the same order first, then workgroups that own 4 rays each run coarse MLP -> compositing ->
    # fine depths -> fine MLP -> compositing (csrc/mlp_render_kernel.h; bit-identical to the launches below)
    if (dev.type == "cuda" and _fusable(models, embeddings) and not (torch.is_grad_enabled() and any(_needs_grad(m) for m in models[:2]))
            and (N_importance == 0 or models[0].mlp_dtype == models[1].mlp_dtype)
            and (not test_time or (FUSE_TEST_TIME and N_importance > 0))
            and ops.render_supported(N_rays, N_samples, N_importance, model_coarse.mlp_dtype)):
        graph_rng = D.in_graph_stream(dev)
        rnd = (lambda *sh: D.rand(sh, dev)) if graph_rng else (lambda *sh: torch.rand(*sh, device=dev))
        rndn = (lambda *sh: D.randn(sh, dev)) if graph_rng else (lambda *sh: torch.randn(*sh, device=dev))
        perturb_rand = rnd(N_rays, N_samples) if perturb > 0 else None                     # :203
        noise_c = rndn(N_rays, N_samples)                                                  # :152 (always drawn)
        u = noise_f = None
        if N_importance > 0:
            u = rnd(N_rays, N_importance) if perturb != 0 else None                        # :39
            noise_f = rndn(N_rays, N_samples + N_importance)                               # :152
        dtype = model_coarse.mlp_dtype
        out = ops.render_fwd(rays, N_samples, N_importance, model_coarse.packed_weights(dtype),
                             models[1].packed_weights(dtype) if N_importance > 0 else None, dtype, use_disp, perturb, perturb_rand,
                             noise_c, noise_f, noise_std, white_back, u, want_coarse=not test_time, test_time=test_time)
        result = {'opacity_coarse': out['opacity_coarse']} if test_time else \
            {'rgb_coarse': out['rgb_coarse'], 'depth_coarse': out['depth_coarse'], 'opacity_coarse': out['opacity_coarse']}
        if N_importance > 0:
            result.update(rgb_fine=out['rgb_fine'], depth_fine=out['depth_fine'], opacity_fine=out['opacity_fine'])
        return result

    # RNG: identical calls, order, shapes and device as the reference.  Inside a hipGraph capture that owns a
    # device-resident generator state (system.GraphedTrainStep: its batch source draws through draws.py) these four come from the
    # same state — torch's own capture-time bookkeeping would restart every replay at the offset that state starts from.
    graph_rng = dev.type == "cuda" and D.in_graph_stream(dev)

    def rand(*shape):
        return D.rand(shape, dev) if graph_rng else torch.rand(*shape, device=dev)

    def randn(*shape):
        return D.randn(shape, dev) if graph_rng else torch.randn(*shape, device=dev)
    perturb_rand = rand(N_rays, N_samples) if perturb > 0 else None                        # :203
    z_vals = ops.sample_coarse_z(rays, N_samples, use_disp, perturb, perturb_rand)          # :189-204
    noise_c = randn(N_rays, N_samples)                                                      # :152 (always drawn)

    raw_c = mlp(model_coarse, z_vals, bool(test_time))                                      # :206-217
    if test_time:
        weights_coarse, opacity_c = ops.composite(raw_c, z_vals, rays, noise_c, noise_std, white_back)
        result = {'opacity_coarse': opacity_c}
    else:
        weights_coarse, opacity_c, rgb_c, depth_c = ops.composite(raw_c, z_vals, rays, noise_c, noise_std, white_back)
        result = {'rgb_coarse': rgb_c, 'depth_coarse': depth_c, 'opacity_coarse': opacity_c}

    if N_importance > 0:                                                                    # :222-242
        u = rand(N_rays, N_importance) if perturb != 0 else None                            # :39, det=(perturb==0)
        z_fine = ops.fine_z(z_vals, weights_coarse.detach(), N_importance, u=u)             # :223-229 (.detach :226)
        noise_f = randn(N_rays, N_samples + N_importance)                                   # :152
        raw_f = mlp(models[1], z_fine, False)
        _, opacity_f, rgb_f, depth_f = ops.composite(raw_f, z_fine, rays, noise_f, noise_std, white_back)
        result['rgb_fine'] = rgb_f
        result['depth_fine'] = depth_f
        result['opacity_fine'] = opacity_f

    return result
