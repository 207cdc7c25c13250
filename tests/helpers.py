import torch

from oracle import nerf_oracle as O


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


def case_from_golden(golden, name, prefix="rr"):
    key = f"{prefix}_{name}_cfg" if prefix == "rr" else f"{prefix}_cfg"
    cfg = golden[key].tolist()
    kind = {0: "blender", 1: "ndc"}[int(cfg[0])]
    B, S_c, N_i = int(cfg[1]), int(cfg[2]), int(cfg[3])
    kw = dict(N_samples=S_c, use_disp=bool(cfg[4]), perturb=cfg[5], noise_std=cfg[6], N_importance=N_i,
              white_back=bool(cfg[7]), test_time=bool(cfg[8]))
    sg, sb, seed = cfg[9], cfg[10], int(cfg[11])
    params = [O.make_params(seed, sg, sb), O.make_params(seed + 500, sg, sb)]
    rays = O.make_rays(seed, B, kind)
    rng = O.draw_rng(seed, B, S_c, N_i, cfg[5])
    return params, rays, kw, rng


def build_models(params, device, dtype="fp32"):
    from nerf_pl_amd.models import Embedding, NeRF
    ms = []
    for p in params:
        m = NeRF()
        m.load_state_dict(p)
        m.mlp_dtype = dtype
        ms.append(m.to(device))
    return ms, [Embedding(3, 10), Embedding(3, 4)]


def hip_render(models, embeddings, rays, kw, rng, device):
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


# ---------------------------------------------------------------------------------------------------------------------
# Procedural "easy" scene for PSNR gates (no dataset offline): a soft-edged ball of smoothly varying colour in front of a
# white background, rendered in closed form (dense quadrature of the analytic field, fp64) — so the targets come from
# neither the HIP path nor the oracle.  A NeRF reaches > 25 dB on it within a few hundred 1024-ray steps.
def analytic_field(x):
    """x (...,3) -> sigma (...), rgb (...,3)."""
    r = x.norm(dim=-1)
    sigma = 40.0 * torch.sigmoid((0.9 - r) * 10.0)
    rgb = 0.5 + 0.4 * torch.stack([torch.sin(1.5 * x[..., 0]), torch.sin(1.5 * x[..., 1] + 1.0),
                                   torch.sin(1.5 * x[..., 2] + 2.0)], -1)
    return sigma, rgb


def analytic_scene(n, seed, device, n_quad=384):
    """n Blender-style rays [o d near=2 far=6] aimed at the ball from a radius-4 sphere + their ground-truth colours."""
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 4.0
    tgt = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * (1.3 * torch.rand(n, 1, generator=g))
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    rays = torch.cat([o, d, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0)], 1).float().contiguous().to(device)
    out = []
    for i in range(0, n, 65536):
        r = rays[i:i + 65536].double()
        t = torch.linspace(2.0, 6.0, n_quad, device=device, dtype=torch.float64)
        pts = r[:, None, :3] + r[:, None, 3:6] * t[None, :, None]
        sigma, rgb = analytic_field(pts)
        delta = (t[1] - t[0]).expand_as(sigma)
        alpha = 1.0 - torch.exp(-sigma * delta)
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], 1), 1)[:, :-1]
        w = alpha * T
        out.append(((w[..., None] * rgb).sum(1) + (1.0 - w.sum(1, keepdim=True))).float())
    return rays, torch.cat(out, 0)
