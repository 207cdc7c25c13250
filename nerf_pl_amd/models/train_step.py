"""The training step's render + loss as ONE autograd node ("fused loss + PSNR + composite backward seed").

`render_rays` (models/rendering.py) stays the modular, drop-in boundary: every operator an autograd node of its own, any loss
on top.  `NeRFSystem.training_step` (train.py:103-117) however always composes the same graph — render_rays -> MSELoss ->
backward -> Adam — and at ~1 ms per step the ~30 small launches and node hops of the modular form are 10-15 % of it.  This
module runs that exact computation with the launches a fixed recipe allows:

    forward   1 launch   the reference's four draws (rendering.py:203, :152, :39, :152) from torch's generator stream — the
                         values torch.rand / randn would return, the generator advanced identically   nerfhip_torch_draws
                         (none at all when the batch brought them along: RayStore.sample(step_draws=...) draws the batch's
                         pixels, generates its rays and makes these draws in ONE launch)
              1 launch   both models' weight images (forward + W^T)                     nerfhip_mlp_pack_weights_train_multi
              coarse     fused MLP forward (saving) that forms its own depths in the prologue     nerfhip_mlp_fwd_rays_coarse
                         compositing + d MSE / d rgb + compositing backward + the fine pass's depths  nerfhip_composite_train_fine_z
              fine       fused MLP forward (saving)                                      nerfhip_mlp_fwd_rays
                         compositing + d MSE / d rgb + compositing backward + loss value and PSNR  nerfhip_composite_train_loss
    backward  per model its chain kernel, then ONE weight-gradient launch and ONE reduce launch for BOTH models
              (optionally with the Adam update applied in the reduce)                    nerfhip_mlp_bwd_multi
 = 10 launches + Adam (round 3: 18 + Adam, round 2: ~32).

Every kernel forms its values with the same expressions as the modular path, so loss, outputs and d loss / d raw are
bit-identical to it; the parameter gradients differ only in the fp32 summation order of the split-K partials.
"""
import os

import numpy as np
import torch

from .. import ops


def fusable(models, embeddings, loss_mod):
    from ..losses import MSELoss
    from .rendering import _fusable
    return (isinstance(loss_mod, MSELoss) and _fusable(models, embeddings)
            and len({m.mlp_dtype for m in models}) == 1
            and all(p.requires_grad for m in models for p in m.parameters()))


def _forward_launches(models, rays, rgbs, S, N, dtype, use_disp, perturb, perturb_rand, noise_c, noise_f, noise_std, white_back, u, gscale,
                      packs, acts_c, dev):
    """The step's forward as four launches (shapes the single-launch kernel does not take; the form it is pinned against)."""
    z, raw_c = ops.mlp_fwd_rays_coarse(rays, S, packs[0][0], False, dtype, use_disp, perturb, perturb_rand, save=acts_c)  # :189-207
    if N > 0:
        _, opac_c, rgb_c, depth_c, g_raw_c, zf = ops.composite_train_fine_z(raw_c, z, rays, noise_c, noise_std, white_back, rgbs,
                                                                            gscale, N, u=u)                   # :143-172, :223-229
        acts_f = ops.alloc_acts(zf.numel(), dtype, dev)
        raw_f = ops.mlp_fwd_rays(rays, zf, packs[1][0], False, dtype, save=acts_f)
        opac_f, rgb_f, depth_f, g_raw_f, out3 = ops.composite_train_loss(raw_f, zf, rays, noise_f, noise_std, white_back, rgbs,
                                                                        gscale, rgb_coarse=rgb_c)   # + losses.py:9-14, metrics.py:4-13
        entries = [(g_raw_f, raw_f, packs[1][1], acts_f), (g_raw_c, raw_c, packs[0][1], acts_c)]   # fine model first (as autograd would)
        outs = [rgb_c, depth_c, opac_c, rgb_f, depth_f, opac_f]
    else:
        opac_c, rgb_c, depth_c, g_raw_c, out3 = ops.composite_train_loss(raw_c, z, rays, noise_c, noise_std, white_back, rgbs, gscale)
        entries = [(g_raw_c, raw_c, packs[0][1], acts_c)]
        outs = [rgb_c, depth_c, opac_c]
    return outs, entries, out3


def _tracked(model):
    """the model's weights are updated by a LIVE optimizer that announces its updates (optim.FlatAdam marks the model with a weak
    reference to itself); load_state_dict bumps the serial on its own (models/nerf.py)"""
    ref = getattr(model, "_serial_tracked", None)
    return ref is not None and ref() is not None


_regen_enc = os.environ.get("NERFHIP_REGEN_ENC", "0") == "1"


def regen_enc_enabled():
    return _regen_enc


def set_regen_enc(on):
    """Process-wide: whether the bf16 fused step leaves the input encodings out of the saved activations and has the weight-gradient
    launch form them again (nerfhip_render_args.regen_enc + nerfhip_mlp_bwd_multi_rays).  Bit-identical gradients; OFF by default
    (NERFHIP_REGEN_ENC=1): measured on one box, alternating, the forward gains 8-10 us in the step and the dW launch — whose ring
    iterations are paced by their instructions, not by their bytes — loses 12-25 (profiles/r06_regen_enc_abab.txt).  Returns the
    previous setting."""
    global _regen_enc
    prev, _regen_enc = _regen_enc, bool(on)
    return prev


class _TrainRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, rays, rgbs, *params):
        models, S, use_disp, perturb, noise_std, N, white_back, adam, draws, packed = cfg
        rays = rays.float().contiguous()
        rgbs = rgbs.float().contiguous()
        B = rays.shape[0]
        dev = rays.device
        dtype = models[0].mlp_dtype
        # RNG: the reference's four draws, same order / shapes / generator stream, one launch — or the tensors the
        # caller hands in (a batch that drew them together with its pixels; a test replaying the reference's recorded draws)
        if draws is None:
            from .. import draws as D
            draws = D.step_draws(B, S, N, perturb, noise_std, dev)
        perturb_rand = draws.get("perturb_rand") if perturb > 0 else None                  # rendering.py:203
        noise_c, noise_f = draws.get("noise_coarse"), draws.get("noise_fine")              # :152 (read only when noise_std != 0)
        u = draws.get("u") if (N > 0 and perturb != 0) else None                           # :39
        if noise_std != 0 and (noise_c is None or (N > 0 and noise_f is None)):
            raise ValueError("render_rays_train: noise_std != 0 needs the noise draws")
        # weight images: the batch's launch packed them (RayStore.sample(pack_models=...)) and the weights have not moved since —
        # else ONE pack launch for both models here
        # (only models whose optimizer ANNOUNCES its updates — FlatAdam bumps `_weights_serial` and marks `_serial_tracked` — can
        # vouch for a pack made before this call; under any other optimizer the serials never move and the images are re-packed)
        fresh = (packed is not None and packed[0] == tuple(id(m) for m in models) and packed[1] == dtype
                 and all(_tracked(m) and getattr(m, "_weights_serial", 0) == sr
                         and getattr(m, "_packed_serial", None) == sr for m, sr in zip(models, packed[2])))
        packs = [m.train_buffers(dtype, dev) for m in models] if fresh else ops.pack_models_train(models, dtype)
        # d mean((rgb - t)^2) / d rgb = (rgb - t) * (2 / n), the quotient formed in fp32 like nerfhip_mse_psnr's `2.0f / (float)n`
        gscale = float(np.float32(2.0) / np.float32(3 * B))
        acts_c = ops.alloc_acts(B * S, dtype, dev)
        if ops.render_supported(B, S, N, dtype):
            # the whole forward in ONE launch (nerfhip_render_train_fwd, csrc/mlp_render_kernel.h): workgroups that own 4 rays each run
            # coarse MLP -> compositing + loss gradient + compositing backward + fine depths -> fine MLP -> the same + loss values
            acts_f = ops.alloc_acts(B * (S + N), dtype, dev) if N > 0 else None
            # bf16, opt-in (set_regen_enc): the positional encodings are not saved (6 of a tile's 151 KiB written, 10 of the 282 KiB
            # the weight-gradient launch reads): that launch forms them again from the rays and the depths, bit for bit
            regen = regen_enc_enabled() and ops.mlp_dtype_code(dtype) == ops.BF16 and S % 32 == 0 and (S + N) % 32 == 0
            o = ops.render_train_fwd(rays, rgbs, gscale, S, N, packs[0][0], packs[1][0] if N > 0 else None, dtype, acts_c, acts_f,
                                     use_disp, perturb, perturb_rand, noise_c, noise_f, noise_std, white_back, u, regen_enc=regen)
            out3 = o["out3"]
            outs = [o["rgb_coarse"], o["depth_coarse"], o["opacity_coarse"]]
            entries = [(o["g_raw_coarse"], o["raw_coarse"], packs[0][1], acts_c) + ((rays, o["z_coarse"]) if regen else ())]
            if N > 0:
                outs += [o["rgb_fine"], o["depth_fine"], o["opacity_fine"]]
                # fine model first (as autograd would)
                entries.insert(0, (o["g_raw_fine"], o["raw_fine"], packs[1][1], acts_f) + ((rays, o["z_fine"]) if regen else ()))
        else:
            outs, entries, out3 = _forward_launches(models, rays, rgbs, S, N, dtype, use_disp, perturb, perturb_rand, noise_c, noise_f, noise_std,
                                                    white_back, u, gscale, packs, acts_c, dev)
        ctx.models = [models[1], models[0]] if N > 0 else [models[0]]
        ctx.entries, ctx.dtype, ctx.adam = entries, dtype, adam
        ctx.n_params = [len(m.flat_params()) for m in models]
        ctx.serials = [m._packed_serial for m in ctx.models]
        ctx.mark_non_differentiable(out3, *outs)
        ctx.set_materialize_grads(False)
        return (out3[0], out3) + tuple(outs)

    @staticmethod
    def backward(ctx, g_loss, *_unused):
        n_models = len(ctx.n_params)
        if g_loss is None:
            return (None, None, None) + (None,) * sum(ctx.n_params)
        entries = ctx.entries
        # d L / d loss multiplies every g_out INSIDE the chain kernels (a device scalar): no scaling launches, whatever the
        # caller passed to backward()
        g_scale = None if ops.is_unit_seed(g_loss) else g_loss.reshape(1).float().contiguous()
        models, dtype = ctx.models, ctx.dtype
        for m, serial in zip(models, ctx.serials):
            m.check_pack_serial(serial)
        hooked = any(getattr(m, "_grad_ready_hook", None) is not None for m in models)
        joint_hook = getattr(models[0], "_grads_ready_hook", None) if hooked else None
        if hooked and joint_hook is not None and all(getattr(m, "_grads_ready_hook", None) == joint_hook for m in models):
            # N > 1 ranks, the default form (parallel.GradSync(form="merged")): the SAME launches as the one-rank step — one chain,
            # one dW, one reduce launch for both models — and then ONE all-reduce over the step's gradients, which the launch wrote as
            # consecutive slices of one buffer.  (Measured at world 1: the per-model form below costs 7.6 % of the step before any
            # wire time — two dW launches fill the chip worse than one; a single 4.77 MB all-reduce exposes less than that.)
            grads = ops.mlp_bwd_multi(entries, dtype, g_scale=g_scale)
            for m, g in zip(models, grads):
                m._flat_grad = g[2]
            joint_hook(models, [g[2] for g in grads])
        elif hooked:
            # N > 1 ranks, form="per_model": per model chain -> dW -> reduce -> grad-ready hook, so that the fine model's all-reduce
            # travels while the coarse model's backward still runs (parallel.GradSync)
            grads = []
            for m, entry in zip(models, entries):
                ((gw, gb, flat),) = ops.mlp_bwd_multi([entry], dtype, g_scale=g_scale)
                m._flat_grad = flat
                m._grad_ready_hook(m, flat)
                grads.append((gw, gb, flat))
        else:
            # Adam inside the reduce kernel only when these gradients ARE the step's gradients: with gradients already accumulated
            # in p.grad (a second backward before the optimizer step) the update would be applied once per backward
            if ctx.adam is not None and any(p.grad is not None for m in models for p in m.flat_params()):
                ctx.adam = None
            adam = ctx.adam.handle(models) if ctx.adam is not None else None
            grads = ops.mlp_bwd_multi(entries, dtype, adam=adam, g_scale=g_scale)
            for m, g in zip(models, grads):
                m._flat_grad = g[2]
            if ctx.adam is not None:
                ctx.adam.applied_in_backward(models)
        ctx.entries = None
        by_model = {id(m): g for m, g in zip(models, grads)}
        out = []
        order = [models[-1]] + models[:-1] if n_models == 2 else models        # parameter order of forward(): coarse, fine
        for m in order:
            gw, gb, _ = by_model[id(m)]
            out += gw + gb
        return (None, None, None) + tuple(out)


def render_rays_train(models, embeddings, rays, rgbs, N_samples=64, use_disp=False, perturb=0, noise_std=1, N_importance=0,
                      white_back=False, adam=None, draws=None, packed=None):
    """Training-mode `render_rays` + MSELoss + PSNR for one ray chunk.  Returns (results, loss, out3): `results` has the keys
    of render_rays (rendering.py:213-244; detached values: the only differentiable output is `loss`, whose backward produces
    the gradients of every parameter of `models`), out3 = [loss, psnr, mse] detached.
    adam: an optim.FlatAdam to apply inside the backward's reduce kernel (single-GPU steps; `optimizer.step()` then skips).
    draws: {'perturb_rand', 'noise_coarse', 'u', 'noise_fine'} tensors to consume instead of drawing (draws.step_specs names the
    shapes): what RayStore.sample(step_draws=...) put into the batch, or the reference's recorded draws in a parity test.
    packed: batch['packed'] of RayStore.sample(pack_models=...) — the weight images are already packed (checked for freshness)."""
    N = int(N_importance)
    use = list(models[:2]) if N > 0 else [models[0]]
    params = [p for m in use for p in m.flat_params()]
    cfg = (use, int(N_samples), bool(use_disp), float(perturb), float(noise_std), N, bool(white_back), adam, draws, packed)
    res = _TrainRender.apply(cfg, rays, rgbs, *params)
    loss, out3 = res[0], res[1]
    results = {'rgb_coarse': res[2], 'depth_coarse': res[3], 'opacity_coarse': res[4]}
    if N > 0:
        results.update(rgb_fine=res[5], depth_fine=res[6], opacity_fine=res[7])
    return results, loss, out3
