"""Loader for the REAL reference hot path (test infrastructure only).

Imports `/root/reference/models/{nerf,rendering}.py` *unmodified*.  The only
thing missing in this image is the un-vendored `torchsearchsorted` submodule
(reference `.gitmodules:1-3`, import at `models/rendering.py:2`, single call
site `models/rendering.py:42` with side='right'); we inject a stand-in with
numpy `searchsorted(side=...)` row-wise semantics built on `torch.searchsorted`.

This module is used ONLY by `oracle/make_golden.py` (to mint fixtures) and by
`tests/test_oracle_vs_reference.py` (skipped when /root/reference is absent, as
on the GPU box).  Nothing in the product path imports it.
"""
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("NERF_PL_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "rendering.py"))


def _install_searchsorted_shim():
    import torch
    if "torchsearchsorted" in sys.modules:
        return
    mod = types.ModuleType("torchsearchsorted")

    def searchsorted(a, v, out=None, side="left"):
        res = torch.searchsorted(a.contiguous(), v.contiguous(), right=(side == "right"))
        if out is not None:
            out.copy_(res)
            return out
        return res

    mod.searchsorted = searchsorted
    sys.modules["torchsearchsorted"] = mod


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def load_reference_ray_utils():
    """`datasets/ray_utils.py` of the reference, unmodified; its only missing import is kornia.create_meshgrid
    (pixel-coordinate grid, x = column, y = row, fp32), replaced by a 5-line stand-in."""
    import torch
    if "ray_utils" not in _cache:
        if "kornia" not in sys.modules:
            kornia = types.ModuleType("kornia")

            def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
                assert not normalized_coordinates
                ys, xs = torch.meshgrid(torch.arange(height, dtype=dtype), torch.arange(width, dtype=dtype), indexing="ij")
                return torch.stack([xs, ys], -1)[None]

            kornia.create_meshgrid = create_meshgrid
            sys.modules["kornia"] = kornia
        _cache["ray_utils"] = _load("_ref_ray_utils", "datasets/ray_utils.py")
    return _cache["ray_utils"]


def load_reference():
    """Returns (ref_nerf_module, ref_rendering_module) loaded from the reference tree
    under private module names (so they never shadow our own `models` package)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if "mods" not in _cache:
        _install_searchsorted_shim()
        nerf = _load("_ref_models_nerf", "models/nerf.py")
        rendering = _load("_ref_models_rendering", "models/rendering.py")
        _cache["mods"] = (nerf, rendering)
    return _cache["mods"]
