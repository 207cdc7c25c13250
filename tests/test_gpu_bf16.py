"""GPU: the gate that licenses the bf16 (and fp8-storage) MFMA configurations — which cannot meet the 1e-4 fp32 parity
bound — as BASELINE.json words it: PSNR within 0.1 dB of the fp32 path at equal steps, plus per-tensor gradient
direction at n >= 1000 points.

The scene is procedural (tests/helpers.analytic_scene: closed-form colours of a soft ball, independent of both the HIP
path and the oracle) and easy enough to pass 25 dB within a few hundred steps, where PSNR is still sensitive.
Every configuration starts from the same default-init weights and consumes the same ray batches and RNG draws."""
from argparse import Namespace

import pytest
import torch

from oracle import nerf_oracle as O
from tests.helpers import analytic_scene, build_models

pytestmark = pytest.mark.gpu

STEPS, B, S, N = 600, 1024, 64, 64
EVAL_AT = (200, 400, 600)


def _train(dtype, dev, rays, rgbs, rays_val, rgb_val, init):
    from nerf_pl_amd.inference import batched_inference
    from nerf_pl_amd.system import NeRFSystem
    hp = Namespace(N_samples=S, N_importance=N, use_disp=False, perturb=1.0, noise_std=0.0, chunk=32768, loss_type="mse",
                   lr=5e-4, weight_decay=0, decay_step=[10 ** 9], decay_gamma=0.5, white_back=True, optimizer="adam",
                   lr_scheduler="steplr")
    system = NeRFSystem(hp)
    system.nerf_coarse.load_state_dict(init[0])
    system.nerf_fine.load_state_dict(init[1])
    for m in system.models:
        m.mlp_dtype = dtype
    system = system.to(dev)
    (opt,), _ = system.configure_optimizers()
    torch.manual_seed(1234)                                  # same perturb / u draws in every configuration
    perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
    curve = {}
    for step in range(1, STEPS + 1):
        idx = perm[((step - 1) * B) % (rays.shape[0] - B):][:B]
        out = system.training_step({"rays": rays[idx], "rgbs": rgbs[idx]}, step)
        opt.zero_grad(set_to_none=True)
        out["loss"].backward()
        opt.step()
        if step in EVAL_AT:
            with torch.no_grad():
                img = batched_inference(system.models, system.embeddings, rays_val, S, N, False, 32768, True)["rgb_fine"]
            curve[step] = (-10 * torch.log10(torch.mean((img - rgb_val) ** 2))).item()
    return curve


def _dtypes():
    from nerf_pl_amd.ops import _DTYPES
    return ["fp32", "bf16"] + (["bf16_f8"] if "bf16_f8" in _DTYPES else [])


def test_psnr_at_equal_steps_within_0p1_db_of_fp32(dev):
    rays, rgbs = analytic_scene(200000, 1, dev)
    rays_val, rgb_val = analytic_scene(8192, 2, dev)
    from nerf_pl_amd.models import NeRF
    torch.manual_seed(0)
    init = [NeRF().state_dict(), NeRF().state_dict()]       # default nn.Linear init, coarse then fine (train.py:38-42)
    curves = {dt: _train(dt, dev, rays, rgbs, rays_val, rgb_val, init) for dt in _dtypes()}
    print("PSNR@step on the analytic scene:", {k: {s: round(v, 3) for s, v in c.items()} for k, c in curves.items()})
    assert curves["fp32"][STEPS] >= 25.0, curves["fp32"]   # the scene is in the PSNR-sensitive regime
    for dt, c in curves.items():
        for s in EVAL_AT:
            assert abs(c[s] - curves["fp32"][s]) <= 0.1, (dt, s, c[s], curves["fp32"][s])


@pytest.mark.parametrize("n", [1000, 4096])
def test_reduced_precision_gradient_direction(dev, n):
    """Per-tensor cosine >= 0.99 between the bf16 (and fp8-storage) gradients and autograd through the fp32 CPU oracle."""
    g = torch.Generator().manual_seed(n)
    p = O.make_params(21, 3.0, 0.1)
    pts = torch.rand(n, 3, generator=g) * 4 - 2
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    x = torch.cat([O.posenc(pts, 10), O.posenc(dirs, 4)], 1)
    g_out = torch.randn(n, 4, generator=g)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    (O.mlp_forward(pr, x) * g_out).sum().backward()
    worst = {}
    for dt in _dtypes()[1:]:
        (m,), _ = build_models([p], dev, dt)
        (m(x.to(dev)) * g_out.to(dev)).sum().backward()
        for name, prm in m.named_parameters():
            ref = pr[name].grad
            cos = torch.nn.functional.cosine_similarity(prm.grad.cpu().flatten(), ref.flatten(), dim=0).item()
            rel = (prm.grad.cpu() - ref).norm().item() / (ref.norm().item() + 1e-12)
            worst[dt] = min(worst.get(dt, 1.0), cos)
            assert cos >= 0.99, (dt, n, name, cos, rel)
    print("worst per-tensor gradient cosine at n=%d:" % n, worst)
