"""Procedural scenes with closed-form ground truth for the PSNR@step gates.  TEST INFRASTRUCTURE ONLY.

No dataset is available offline (BASELINE.json: Blender/lego), so the PSNR half of the metric is measured on fields
whose pixel colours are computed in closed form (dense fp64 quadrature of an analytic density / colour field) — i.e. by
neither the HIP path nor the oracle.

* `analytic_scene`  — a soft-edged ball of smoothly varying colour: a NeRF passes 40 dB within ~1000 steps.  The "easy" scene
  of rounds 1-2; at 42 dB the bf16 rounding of the network itself is a visible part of the residual.
* `brick_scene`     — "lego-like": a box + a ball with sharp density edges carrying a hard-edged checker texture the network
  learns within the step budget, plus a fine grain far above the bandwidth of the positional encoding (detail no 8x256
  NeRF can represent: what makes a real scene's PSNR@step PLATEAU).  With the default parameters the grain alone bounds
  PSNR at ~32.6 dB and fp32 training ends at 31.3 dB — the reference's lego range (README.md:161: 31.39 dB;
  test.ipynb:132: 30.65 dB), the regime in which `north_star` asks for "PSNR within 0.1 dB of the reference at equal steps".
  `grain=0` removes the plateau: the same recipe is then still climbing at 30-31 dB and trajectory chaos (+-0.6 dB per run
  pair) swamps any arithmetic effect (profiles/archive/r03_psnr_gate_brick_no_grain.json).
"""
import torch


def _render_closed_form(field, rays, n_quad, chunk=32768):
    """Quadrature of `field` along Blender-style rays [o d near far] with white background, fp64."""
    out = []
    for i in range(0, rays.shape[0], chunk):
        r = rays[i:i + chunk].double()
        t = torch.linspace(0.0, 1.0, n_quad, device=r.device, dtype=torch.float64)
        t = r[:, 6:7] * (1.0 - t) + r[:, 7:8] * t
        pts = r[:, None, :3] + r[:, None, 3:6] * t[..., None]
        sigma, rgb = field(pts)
        delta = (t[:, 1:2] - t[:, 0:1]).expand_as(sigma)
        alpha = 1.0 - torch.exp(-sigma * delta)
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], 1), 1)[:, :-1]
        w = alpha * T
        out.append(((w[..., None] * rgb).sum(1) + (1.0 - w.sum(1, keepdim=True))).float())
    return torch.cat(out, 0)


def _aimed_rays(n, seed, device, spread):
    """n rays from a radius-4 sphere aimed at random points within `spread` of the origin; near 2, far 6 (blender.py:34-35)."""
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 4.0
    tgt = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * (spread * torch.rand(n, 1, generator=g))
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    return torch.cat([o, d, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0)], 1).float().contiguous().to(device)


# ---------------------------------------------------------------------------------------------------------- easy scene
def analytic_field(x):
    """x (...,3) -> sigma (...), rgb (...,3)."""
    r = x.norm(dim=-1)
    sigma = 40.0 * torch.sigmoid((0.9 - r) * 10.0)
    rgb = 0.5 + 0.4 * torch.stack([torch.sin(1.5 * x[..., 0]), torch.sin(1.5 * x[..., 1] + 1.0),
                                   torch.sin(1.5 * x[..., 2] + 2.0)], -1)
    return sigma, rgb


def analytic_scene(n, seed, device, n_quad=384):
    """n Blender-style rays [o d near=2 far=6] aimed at the ball from a radius-4 sphere + their ground-truth colours."""
    rays = _aimed_rays(n, seed, device, 1.3)
    return rays, _render_closed_form(analytic_field, rays, n_quad, chunk=65536)


# ----------------------------------------------------------------------------------------------------- lego-like scene
BRICK_DEFAULT = dict(freq=6.0, amp=0.2, sharp=2.0, edge=30.0, grain=0.45)


def brick_field(x, freq=6.0, amp=0.2, sharp=2.0, edge=30.0, grain=0.45):
    """A 1.4 x 1.0 x 0.7 box with a radius-0.45 ball sitting on it: density 60 inside, edges `edge` per unit; colour = a
    smooth base + `amp` x a hard-edged 3-D checker (tanh(sharp * product of sines at `freq` rad per unit))."""
    q = x.abs() - torch.tensor([0.7, 0.5, 0.35], dtype=x.dtype, device=x.device)
    box = q.clamp(min=0).norm(dim=-1) + q.max(dim=-1).values.clamp(max=0)                  # signed distance, box
    ball = (x - torch.tensor([0.15, 0.1, 0.55], dtype=x.dtype, device=x.device)).norm(dim=-1) - 0.45
    sdf = torch.minimum(box, ball)
    sigma = 60.0 * torch.sigmoid(-edge * sdf)
    s = torch.sin(freq * x[..., 0] + 0.3) * torch.sin(freq * x[..., 1] + 1.1) * torch.sin(freq * x[..., 2] + 2.3)
    checker = torch.tanh(sharp * s)
    base = 0.5 + 0.08 * torch.stack([torch.sin(2.0 * x[..., 0]), torch.sin(2.0 * x[..., 1] + 1.0), torch.sin(2.0 * x[..., 2] + 2.0)], -1)
    tint = torch.tensor([1.0, -0.6, 0.35], dtype=x.dtype, device=x.device)
    rgb = base + amp * checker[..., None] * tint
    if grain:
        # fine grain far above the bandwidth of the 10-octave positional encoding (2^9 rad per unit): detail no 8x256 NeRF can
        # represent — the part of a real scene's residual that makes PSNR@step plateau instead of rising for ever
        g = torch.sin(1301.0 * x[..., 0] + 0.7) * torch.sin(1487.0 * x[..., 1] + 1.9) * torch.sin(1693.0 * x[..., 2] + 2.9)
        rgb = rgb + grain * g[..., None]
    return sigma, rgb.clamp(0.0, 1.0)


def brick_scene(n, seed, device, n_quad=768, spread=1.4, **params):
    """n rays aimed at the brick (about a third of them miss it: white background) + closed-form colours.
    `params` override BRICK_DEFAULT (freq, amp, sharp, edge)."""
    p = dict(BRICK_DEFAULT)
    p.update(params)
    rays = _aimed_rays(n, seed, device, spread)
    return rays, _render_closed_form(lambda pts: brick_field(pts, **p), rays, n_quad)
