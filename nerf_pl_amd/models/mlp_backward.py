"""Backward of the fused MLP: thin dispatch onto nerfhip_mlp_bwd (chain + dW + reduce kernels)."""
from .. import ops


def _param_grads(model, out, acts, dtype, g_out):
    packed_bwd = model.packed_weights_bwd(dtype)
    gw, gb, flat = ops.mlp_bwd(g_out, out, packed_bwd, acts, dtype)
    model._flat_grad = flat          # contiguous view of this step's gradients (parallel.GradSync uses it)
    need = [p.requires_grad for p in model.flat_params()]
    grads = gw + gb
    return [g if nd else None for g, nd in zip(grads, need)]


def backward_rays(model, out, acts, dtype, g_out):
    return _param_grads(model, out, acts, dtype, g_out)


def backward_embedded(model, out, acts, dtype, g_out):
    return _param_grads(model, out, acts, dtype, g_out)
