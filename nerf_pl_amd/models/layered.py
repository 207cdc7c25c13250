"""NeRF.forward for shapes other than the reference's default: one HIP GEMM launch per nn.Linear (csrc/linear.hip).

The reference's NeRF takes any depth D, width W, skip set and channel counts (models/nerf.py:42-81); its own scripts never
pass anything but the defaults (train.py:38-42, eval.py:100-103), which is what the fused register-resident kernels are built
for.  Every other shape runs here, with the layer graph of nerf.py:100-124: the skip concat `[input_xyz, h]` (nerf.py:108-109)
and the direction concat `[final, input_dir]` (nerf.py:118) are never materialised — a layer reads its two sources through two
column blocks of its weight.  Each layer is an autograd node (ops.linear_act), so parameters and input are differentiable.
"""
import torch

from .. import ops


def nerf_forward(model, x, sigma_only=False):
    """x (n, in_channels_xyz + in_channels_dir) [or (n, in_channels_xyz) when sigma_only] -> (n, 4) = [rgb, sigma] / (n, 1)."""
    dtype = model.mlp_dtype
    c_xyz, c_dir = model.in_channels_xyz, model.in_channels_dir
    need = c_xyz if sigma_only else c_xyz + c_dir
    if x.dim() != 2 or x.shape[1] != need:
        raise ValueError("NeRF.forward expects %d input channels, got %s" % (need, tuple(x.shape)))
    if 0 in model.skips:
        raise ValueError("skips may not contain 0: the first layer has no hidden state to concatenate (nerf.py:61-66)")
    x = x.float()
    xyz = x[:, :c_xyz]
    h = xyz
    for i in range(model.D):
        lin = getattr(model, "xyz_encoding_%d" % (i + 1))[0]
        src = [xyz, h] if i in model.skips else [h]                                   # nerf.py:108-109
        h = ops.linear_act(src, lin.weight, lin.bias, ops.ACT_RELU, dtype)
    sigma = ops.linear_act([h], model.sigma.weight, model.sigma.bias, ops.ACT_NONE, dtype)            # nerf.py:112
    if sigma_only:
        return sigma
    final = ops.linear_act([h], model.xyz_encoding_final.weight, model.xyz_encoding_final.bias, ops.ACT_NONE, dtype)
    d = model.dir_encoding[0]
    t = ops.linear_act([final, x[:, c_xyz:]], d.weight, d.bias, ops.ACT_RELU, dtype)                   # nerf.py:118-119
    rgb = ops.linear_act([t], model.rgb[0].weight, model.rgb[0].bias, ops.ACT_SIGMOID, dtype)
    return torch.cat([rgb, sigma], -1)                                                                 # nerf.py:122
