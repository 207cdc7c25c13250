"""Backward of the fused MLP (K2b).  Filled in by the training milestone."""


def backward_rays(model, rays, z, sigma_only, dtype, g_out):
    raise NotImplementedError("nerf_pl_amd: MLP backward kernels not built yet")


def backward_embedded(model, x, sigma_only, dtype, g_out, need_gx=False):
    raise NotImplementedError("nerf_pl_amd: MLP backward kernels not built yet")
