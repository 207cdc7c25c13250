"""nerf_pl_amd — MI355X-native (gfx950) implementation of the NeRF volume-rendering hot path of
kwea123/nerf_pl: `models/nerf.py` (Embedding, NeRF) + `models/rendering.py` (render_rays, sample_pdf).

Drop-in use from the reference tree:

    import nerf_pl_amd; nerf_pl_amd.install()      # makes `models.nerf` / `models.rendering` /
                                                   # `torchsearchsorted` resolve to this package
    from models.nerf import Embedding, NeRF
    from models.rendering import render_rays

All compute runs in hand-written HIP kernels behind a C ABI (include/nerfhip.h, libnerfhip.so);
there is no CPU or eager fallback.
"""
import os
import sys
import types

_DEFAULT_MLP_DTYPE = os.environ.get("NERF_PL_AMD_MLP_DTYPE", "fp32")


def set_default_mlp_dtype(dtype):
    """'fp32' (exact-fp32 MFMA, the parity configuration) or 'bf16' (bf16 MFMA, fp32 accumulate)."""
    global _DEFAULT_MLP_DTYPE
    from .ops import mlp_dtype_code
    mlp_dtype_code(dtype)
    _DEFAULT_MLP_DTYPE = dtype


def default_mlp_dtype():
    return _DEFAULT_MLP_DTYPE


def install():
    """Register this package's modules under the names the reference imports
    (train.py:10-11, eval.py:9-10, models/rendering.py:2)."""
    from . import models, ops
    from .models import nerf, rendering
    sys.modules["models"] = models
    sys.modules["models.nerf"] = nerf
    sys.modules["models.rendering"] = rendering
    tss = types.ModuleType("torchsearchsorted")
    tss.searchsorted = ops.searchsorted
    sys.modules["torchsearchsorted"] = tss
