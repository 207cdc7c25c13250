"""autograd glue between nn.Module parameters and the fused HIP MLP kernels.

Training forward = the same fused kernel with `save_acts` (every layer's B-operand slabs written once,
in register order); backward = nerfhip_mlp_bwd.  Gradients flow to the 24 parameter tensors only:
the reference detaches the importance samples (rendering.py:226) and its rays carry no grad, so
d/d(rays, z) is never needed; d/dx of pre-embedded inputs is not implemented (raises)."""
import torch

from .. import ops
from . import mlp_backward


def _needs_grad(model):
    return torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())


class _MLPRays(torch.autograd.Function):
    """raw = MLP(posenc(o + d z), posenc(d)) for every sample of every ray (rendering.py:115-141)."""

    @staticmethod
    def forward(ctx, model, rays, z, *params):
        dtype = model.mlp_dtype
        acts = ops.alloc_acts(z.numel(), dtype, z.device)
        out = ops.mlp_fwd_rays(rays, z, model.packed_weights(dtype), False, dtype, save=acts)
        ctx.model, ctx.dtype, ctx.acts = model, dtype, acts
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (out,) = ctx.saved_tensors
        grads = mlp_backward.backward_rays(ctx.model, out, ctx.acts, ctx.dtype, g_out)
        ctx.acts = None
        return (None, None, None) + tuple(grads)


class _MLPEmbedded(torch.autograd.Function):
    """NeRF.forward on pre-embedded inputs (nerf.py:83-124)."""

    @staticmethod
    def forward(ctx, model, x, *params):
        dtype = model.mlp_dtype
        acts = ops.alloc_acts(x.shape[0], dtype, x.device)
        out = ops.mlp_fwd_embedded(x, model.packed_weights(dtype), False, dtype, save=acts)
        ctx.model, ctx.dtype, ctx.acts = model, dtype, acts
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("nerf_pl_amd: gradient w.r.t. pre-embedded NeRF inputs is not implemented "
                                      "(never needed by the reference: rays carry no grad)")
        (out,) = ctx.saved_tensors
        grads = mlp_backward.backward_embedded(ctx.model, out, ctx.acts, ctx.dtype, g_out)
        ctx.acts = None
        return (None, None) + tuple(grads)


def mlp_rays(model, rays, z, sigma_only):
    if _needs_grad(model) and not sigma_only:
        return _MLPRays.apply(model, rays, z, *model.flat_params())
    return ops.mlp_fwd_rays(rays, z, model.packed_weights(), sigma_only, model.mlp_dtype)


def mlp_embedded(model, x, sigma_only):
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).float()
    if (_needs_grad(model) or (torch.is_grad_enabled() and x2.requires_grad)) and not sigma_only:
        out = _MLPEmbedded.apply(model, x2, *model.flat_params())
    else:
        out = ops.mlp_fwd_embedded(x2, model.packed_weights(), sigma_only, model.mlp_dtype)
    return out.reshape(*lead, out.shape[-1])
