"""CPU: the oracle restatement for NON-default NeRF / Embedding shapes against vectors minted from the real reference
(oracle/make_golden_arch.py -> tests/golden/reference_golden_arch.npz): NeRF.forward, its gradients, render_rays and the
gradients of the training loss."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from oracle.arch_cases import ARCHS, N_I, S_C

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_golden_arch.npz")


@pytest.fixture(scope="module")
def garch():
    z = np.load(PATH, allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.mark.parametrize("tag", sorted(ARCHS))
def test_oracle_mlp_and_grads(garch, tag):
    kw, seed = ARCHS[tag]
    arch = O.make_arch(**kw)
    p = O.make_params(seed, 6.0, 0.3, arch=arch)
    for v in p.values():
        v.requires_grad_(True)
    x = garch[tag + "/x"].clone().requires_grad_(True)
    out = O.mlp_forward(p, x, arch=arch)
    assert torch.allclose(out.detach(), garch[tag + "/out"], rtol=1e-5, atol=1e-6)
    (out * garch[tag + "/G"]).sum().backward()
    assert torch.allclose(x.grad, garch[tag + "/gx"], rtol=1e-4, atol=1e-6)
    for n, v in p.items():
        ref = garch[tag + "/g/" + n]
        assert torch.allclose(v.grad, ref, rtol=1e-4, atol=1e-5 * ref.abs().max().item() + 1e-9), n
    with torch.no_grad():
        sig = O.mlp_forward(p, x[:, :arch["in_xyz"]], sigma_only=True, arch=arch)
    assert torch.allclose(sig, garch[tag + "/sigma_only"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", sorted(ARCHS))
def test_oracle_render_and_training_grads(garch, tag):
    kw, seed = ARCHS[tag]
    arch = O.make_arch(**kw)
    params = [O.make_params(seed + k, 6.0, 0.3, arch=arch) for k in range(2)]
    for d in params:
        for v in d.values():
            v.requires_grad_(True)
    rays = garch[tag + "/rays"]
    res = O.render_rays(params, rays, S_C, False, 0, 0, N_I, True, False, arch=arch)
    for k, v in res.items():
        assert torch.allclose(v.detach(), garch[tag + "/render/" + k], rtol=1e-5, atol=1e-6), k
    loss = O.mse_loss(res, garch[tag + "/target"])
    assert abs(loss.item() - garch[tag + "/loss"].item()) <= 1e-5 * abs(loss.item())
    loss.backward()
    for mi, d in enumerate(params):
        for n, v in d.items():
            ref = garch[tag + "/rg%d/" % mi + n]
            assert (O.grad_digest(v.grad) - ref).abs().max().item() <= 2e-4 * (ref[1].abs().item() + 1e-12) + 1e-9, (mi, n)
    with torch.no_grad():
        tt = O.render_rays(params, rays, S_C, False, 0, 0, N_I, True, True, arch=arch)
    assert sorted(tt) == sorted(k[len(tag + "/render_tt/"):] for k in garch if k.startswith(tag + "/render_tt/"))
    for k, v in tt.items():
        assert torch.allclose(v, garch[tag + "/render_tt/" + k], rtol=1e-5, atol=1e-6), k
