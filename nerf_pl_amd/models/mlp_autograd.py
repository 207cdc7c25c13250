"""autograd glue between nn.Module parameters and the fused HIP MLP kernels."""
import torch

from .. import ops


def _needs_grad(model, *tensors):
    if not torch.is_grad_enabled():
        return False
    return any(p.requires_grad for p in model.parameters()) or any(t is not None and t.requires_grad for t in tensors)


class _MLPRays(torch.autograd.Function):
    """raw = MLP(posenc(o + d z), posenc(d)) for every sample of every ray (rendering.py:115-141)."""

    @staticmethod
    def forward(ctx, model, rays, z, sigma_only, *params):
        dtype = model.mlp_dtype
        packed = model.packed_weights(dtype)
        out = ops.mlp_fwd_rays(rays, z, packed, sigma_only, dtype)
        ctx.model, ctx.sigma_only, ctx.dtype = model, sigma_only, dtype
        ctx.save_for_backward(rays, z)
        return out

    @staticmethod
    def backward(ctx, g_out):
        from . import mlp_backward
        rays, z = ctx.saved_tensors
        grads = mlp_backward.backward_rays(ctx.model, rays, z, ctx.sigma_only, ctx.dtype, g_out)
        return (None, None, None, None) + tuple(grads)


class _MLPEmbedded(torch.autograd.Function):
    """NeRF.forward on pre-embedded inputs (nerf.py:83-124)."""

    @staticmethod
    def forward(ctx, model, x, sigma_only, *params):
        dtype = model.mlp_dtype
        packed = model.packed_weights(dtype)
        out = ops.mlp_fwd_embedded(x, packed, sigma_only, dtype)
        ctx.model, ctx.sigma_only, ctx.dtype = model, sigma_only, dtype
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g_out):
        from . import mlp_backward
        (x,) = ctx.saved_tensors
        gx, grads = mlp_backward.backward_embedded(ctx.model, x, ctx.sigma_only, ctx.dtype, g_out,
                                                   need_gx=ctx.needs_input_grad[1])
        return (None, gx, None) + tuple(grads)


def mlp_rays(model, rays, z, sigma_only):
    if _needs_grad(model):
        return _MLPRays.apply(model, rays, z, sigma_only, *model.flat_params())
    return ops.mlp_fwd_rays(rays, z, model.packed_weights(), sigma_only, model.mlp_dtype)


def mlp_embedded(model, x, sigma_only):
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).float()
    if _needs_grad(model, x2):
        out = _MLPEmbedded.apply(model, x2, sigma_only, *model.flat_params())
    else:
        out = ops.mlp_fwd_embedded(x2, model.packed_weights(), sigma_only, model.mlp_dtype)
    return out.reshape(*lead, out.shape[-1])
