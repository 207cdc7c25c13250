"""autograd glue between nn.Module parameters and the fused HIP MLP kernels.

Training forward = the same fused kernel with `save_acts` (every layer's B-operand slabs written once,
in register order); backward = nerfhip_mlp_bwd.  Gradients flow to the 24 parameter tensors only:
the reference detaches the importance samples (rendering.py:226) and its rays carry no grad, so
d/d(rays, z) is never needed; d/dx of pre-embedded inputs (NeRF.forward called directly) comes from
nerfhip_mlp_dx_embedded."""
import torch

from .. import ops


def _param_grads(model, out, acts, dtype, g_out, packed_bwd, workspace=None):
    """nerfhip_mlp_bwd (chain + dW + reduce kernels) -> gradients in `flat_params()` order.  `packed_bwd`: the W^T image the
    forward packed together with its own (NeRF.packed_weights_train)."""
    gw, gb, flat = ops.mlp_bwd(g_out, out, packed_bwd, acts, dtype, workspace=workspace)
    model._flat_grad = flat          # contiguous view of this step's gradients (parallel.GradSync / FlatAdam use it)
    hook = getattr(model, "_grad_ready_hook", None)
    if hook is not None:             # parallel.GradSync: start this model's all-reduce while autograd keeps going
        hook(model, flat)
    need = [p.requires_grad for p in model.flat_params()]
    return [g if nd else None for g, nd in zip(gw + gb, need)]


def _needs_grad(model):
    return torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())


class _MLPRays(torch.autograd.Function):
    """raw = MLP(posenc(o + d z), posenc(d)) for every sample of every ray (rendering.py:115-141)."""

    @staticmethod
    def forward(ctx, model, rays, z, *params):
        dtype = model.mlp_dtype
        acts = ops.alloc_acts(z.numel(), dtype, z.device)
        packed, packed_bwd = model.packed_weights_train(dtype)
        out = ops.mlp_fwd_rays(rays, z, packed, False, dtype, save=acts)
        ctx.model, ctx.dtype, ctx.acts, ctx.packed_bwd = model, dtype, acts, packed_bwd
        ctx.serial = model._packed_serial
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (out,) = ctx.saved_tensors
        ctx.model.check_pack_serial(ctx.serial)
        grads = _param_grads(ctx.model, out, ctx.acts, ctx.dtype, g_out, ctx.packed_bwd)
        ctx.acts = ctx.packed_bwd = None
        return (None, None, None) + tuple(grads)


class _MLPEmbedded(torch.autograd.Function):
    """NeRF.forward on pre-embedded inputs (nerf.py:83-124)."""

    @staticmethod
    def forward(ctx, model, x, *params):
        dtype = model.mlp_dtype
        if dtype == "bf16_f8" and ctx.needs_input_grad[1]:
            # d/dx needs dY of three layers at full bf16 precision (nerfhip_mlp_dx_embedded); the 8-bit mode keeps dY only as
            # e5m2 copies for the dW GEMM.  Same bf16 MFMA forward and chain: this call simply saves its tensors in bf16.
            dtype = "bf16"
        acts = ops.alloc_acts(x.shape[0], dtype, x.device)
        packed, packed_bwd = model.packed_weights_train(dtype)
        out = ops.mlp_fwd_embedded(x, packed, False, dtype, save=acts)
        ctx.model, ctx.dtype, ctx.acts, ctx.packed_bwd = model, dtype, acts, packed_bwd
        ctx.serial = model._packed_serial
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (out,) = ctx.saved_tensors
        ctx.model.check_pack_serial(ctx.serial)
        ws = {} if ctx.needs_input_grad[1] else None
        grads = _param_grads(ctx.model, out, ctx.acts, ctx.dtype, g_out, ctx.packed_bwd, workspace=ws)
        ctx.acts = ctx.packed_bwd = None
        gx = None
        if ctx.needs_input_grad[1]:          # nerf.py:100-124 is differentiable w.r.t. x: dx from the chain's dY slabs
            m = ctx.model
            gx = ops.mlp_dx_embedded(ws["dys"], out.shape[0], m.xyz_encoding_1[0].weight, m.xyz_encoding_5[0].weight,
                                     m.dir_encoding[0].weight, ctx.dtype)
        return (None, gx) + tuple(grads)


def mlp_rays(model, rays, z, sigma_only):
    if _needs_grad(model):
        out = _MLPRays.apply(model, rays, z, *model.flat_params())
        # sigma_only under grad (render_rays(test_time=True) outside no_grad): the reference's sigma-only forward is
        # differentiable (nerf.py:103-114), so evaluate the full network and keep the density channel; the colour
        # branch then simply receives zero gradient.
        return out[..., 3] if sigma_only else out
    return ops.mlp_fwd_rays(rays, z, model.packed_weights(), sigma_only, model.mlp_dtype)


def mlp_embedded(model, x, sigma_only):
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).float()
    if _needs_grad(model) or (torch.is_grad_enabled() and x2.requires_grad):
        if sigma_only:
            # differentiable sigma-only forward (nerf.py:103-114): full network on [xyz | 0 direction channels], density
            # channel kept; colour-branch parameters get zero gradient
            if x2.shape[1] != model.in_channels_xyz:
                raise ValueError("NeRF.forward expects %d input channels, got %d" % (model.in_channels_xyz, x2.shape[1]))
            x2 = torch.cat([x2, x2.new_zeros(x2.shape[0], model.in_channels_dir)], 1)
            out = _MLPEmbedded.apply(model, x2, *model.flat_params())[:, 3:4]
        else:
            out = _MLPEmbedded.apply(model, x2, *model.flat_params())
    else:
        out = ops.mlp_fwd_embedded(x2, model.packed_weights(), sigma_only, model.mlp_dtype)
    return out.reshape(*lead, out.shape[-1])
