"""Replayed randomness for comparing the HIP `render_rays` with the oracle.  TEST INFRASTRUCTURE ONLY.

The reference draws its jitter / noise inside `render_rays` (rendering.py:203, :152, :39, :152); the oracle takes the
draws as injected tensors (`nerf_oracle.draw_rng`).  `hip_render` makes the HIP path consume the same tensors by
standing in for the `torch` name inside `nerf_pl_amd.models.rendering` for the duration of one call.
Used by `tests/` and `__graft_entry__.smoke()`; the product never imports this package."""
import torch


class ReplayRNG:
    """Stand-in for the `torch` name inside nerf_pl_amd.models.rendering: rand/randn return queued
    tensors (moved to the requested device) so the HIP path consumes the same draws as the oracle."""

    def __init__(self, rng, order, device):
        self.q = [(k, rng[k]) for k in order if k in rng]
        self.device = device

    def _pop(self, shape):
        k, t = self.q.pop(0)
        assert tuple(t.shape) == tuple(shape), (k, t.shape, shape)
        return t.to(self.device)

    def rand(self, *shape, **kw):
        return self._pop(shape)

    def randn(self, *shape, **kw):
        return self._pop(shape)

    def __getattr__(self, name):
        return getattr(torch, name)


def hip_render(models, embeddings, rays, kw, rng, device):
    """render_rays of the HIP path on `device` with the oracle's draws `rng` (same order as rendering.py)."""
    from nerf_pl_amd.models import rendering
    kw = dict(kw)
    order = []
    if kw["perturb"] > 0:
        order.append("perturb_rand")
    order.append("noise_coarse")
    if kw["N_importance"] > 0:
        if kw["perturb"] != 0:
            order.append("u")
        order.append("noise_fine")
    replay = ReplayRNG(rng, order, device)
    saved = rendering.torch
    rendering.torch = replay
    try:
        res = rendering.render_rays(models, embeddings, rays.to(device), kw["N_samples"], kw["use_disp"], kw["perturb"],
                                    kw["noise_std"], kw["N_importance"], 1024 * 32, kw["white_back"],
                                    test_time=kw["test_time"])
    finally:
        rendering.torch = saved
    assert not replay.q
    return res


def fused_draws(rng, kw, device):
    """The same recorded draws in the form the FUSED training node consumes (nerf_pl_amd.models.train_step.render_rays_train(...,
    draws=...)): the node that bench.py times then runs on exactly the tensors the reference drew when the golden gradients
    were minted (oracle/make_golden.py), like `hip_render` does for the modular render_rays."""
    keys = []
    if kw["perturb"] > 0:
        keys.append("perturb_rand")
    keys.append("noise_coarse")
    if kw["N_importance"] > 0:
        if kw["perturb"] != 0:
            keys.append("u")
        keys.append("noise_fine")
    return {k: rng[k].to(device).float().contiguous() for k in keys if k in rng}
