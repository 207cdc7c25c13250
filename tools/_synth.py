"""Seeded synthetic inputs for the measurement tools (no dependency on oracle/): Blender-like rays and default-init NeRF weights
with a sharpened density head, the shapes of SURVEY §8(d)."""
import torch


def make_rays(seed, n, device):
    """(n, 8) = [o(3) d(3) near far]: o = (0, 0, 4) + 0.1 N, unit d aimed roughly at the origin, near 2, far 6 (blender.py:34-35)"""
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(n, 3, generator=g)
    d = 0.8 * torch.randn(n, 3, generator=g) - o
    d = d / d.norm(dim=-1, keepdim=True)
    return torch.cat([o, d, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0)], 1).float().contiguous().to(device)


def make_model(seed, device, dtype, sigma_gain=4.0, sigma_bias=0.2):
    """NeRF() with nn.Linear's default init under `seed`, density head rescaled so that opacities saturate like a trained field"""
    from nerf_pl_amd.models import NeRF
    torch.manual_seed(seed)
    m = NeRF()
    with torch.no_grad():
        m.sigma.weight.mul_(sigma_gain)
        m.sigma.bias.mul_(sigma_gain).add_(sigma_bias)
    m.mlp_dtype = dtype
    return m.to(device)


def make_pose(seed):
    """a camera-to-world (3, 4) looking at the origin from radius 4"""
    g = torch.Generator().manual_seed(seed)
    c = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * 4.0
    z = torch.nn.functional.normalize(c, dim=0)
    up = torch.tensor([0.0, 0.0, 1.0])
    x = torch.nn.functional.normalize(torch.linalg.cross(up, z), dim=0)
    y = torch.linalg.cross(z, x)
    return torch.stack([x, y, z, c], 1).float()
