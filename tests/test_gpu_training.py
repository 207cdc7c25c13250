"""GPU: backward kernels (mlp_bwd_chain + mlp_bwd_dw + reduce, composite_bwd) against autograd through
the CPU oracle and against the gradient digests minted from the real reference.

Tolerances: fp32 MFMA path 2e-4 of each tensor's max |grad| (different fp32 summation order over up to
~10^4 points); bf16 path 4e-2 of max |grad| (bf16 activations + bf16 dY, fp32 accumulate) — bf16 is the
roofline configuration, gated on PSNR rather than on gradient bits."""
import pytest
import torch

from oracle import nerf_oracle as O
from tests.helpers import build_models, case_from_golden, hip_render

pytestmark = pytest.mark.gpu


def _oracle_param_grads(p, x, g_out):
    p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    out = O.mlp_forward(p, x)
    (out * g_out).sum().backward()
    return {k: v.grad for k, v in p.items()}, out.detach()


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-4), ("bf16", 4e-2)])
@pytest.mark.parametrize("n", [1, 33, 300, 1000])
def test_mlp_backward_embedded_vs_autograd(dev, dtype, tol, n):
    g = torch.Generator().manual_seed(n)
    p = O.make_params(21, 3.0, 0.1)
    pts = torch.rand(n, 3, generator=g) * 4 - 2
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    x = torch.cat([O.posenc(pts, 10), O.posenc(dirs, 4)], 1)
    g_out = torch.randn(n, 4, generator=g)
    ref, ref_out = _oracle_param_grads(p, x, g_out)
    (m,), _ = build_models([p], dev, dtype)
    out = m(x.to(dev))
    (out * g_out.to(dev)).sum().backward()
    assert (out.detach().cpu() - ref_out).abs().max().item() <= (1e-5 if dtype == "fp32" else 3e-2) * max(1.0, ref_out.abs().max().item())
    for name, prm in m.named_parameters():
        assert prm.grad is not None, name
        r = ref[name]
        gq = prm.grad.cpu()
        if dtype == "fp32":
            err = (gq - r).abs().max().item()
            assert err <= tol * r.abs().max().item() + 1e-7, (dtype, n, name, err, r.abs().max().item())
        else:   # bf16: relative L2 error of the whole tensor (rounding does not average out for tiny n)
            # measured: 0.1-11% relative L2 (deepest layers worst), cosine >= 0.993, because the bf16
            # FORWARD activations (hence ReLU gates) already differ from the fp32 oracle's
            rel = (gq - r).norm().item() / (r.norm().item() + 1e-12)
            cos = torch.nn.functional.cosine_similarity(gq.flatten(), r.flatten(), dim=0).item()
            assert rel <= (0.30 if n < 64 else 0.16) and cos >= (0.95 if n < 64 else 0.985), (dtype, n, name, rel, cos)


def test_training_grads_fp32_vs_reference_golden(golden, dev):
    params, rays, kw, rng = case_from_golden(golden, None, prefix="gr")
    ms, emb = build_models(params, dev, "fp32")
    res = hip_render(ms, emb, rays, kw, rng, dev)
    tgt = golden["gr_target"].to(dev)
    loss = torch.nn.functional.mse_loss(res["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(res["rgb_fine"], tgt)
    loss.backward()
    assert abs(loss.item() - golden["gr_loss"].item()) <= 1e-4 * abs(golden["gr_loss"].item())
    assert torch.allclose(res["rgb_fine"].detach().cpu(), golden["gr_rgb_fine"], rtol=1e-4, atol=1e-4)
    for tag, m in (("c", ms[0]), ("f", ms[1])):
        for n, prm in m.named_parameters():
            dig = O.grad_digest(prm.grad.cpu())
            ref = golden[f"gr_{tag}_{n}"]
            scale = ref[1].abs().item() + 1e-12          # l2 norm of the reference gradient tensor
            # digest = [sum, l2, 8 head, 8 tail]; the signed sum of up to 81k elements cancels heavily,
            # so it is compared against the tensor's l2 norm
            # (|sum| can exceed l2 by up to sqrt(numel), so each entry also gets its own 2e-3 rel band)
            assert bool(((dig - ref).abs() <= 4e-3 * scale + 2e-3 * ref.abs() + 1e-9).all()), (tag, n, dig[:4], ref[:4])
    assert torch.allclose(ms[0].sigma.weight.grad.cpu(), golden["gr_full_c_sigma.weight"], rtol=2e-3, atol=1e-7)
    assert torch.allclose(ms[1].rgb[0].weight.grad.cpu(), golden["gr_full_f_rgb.0.weight"], rtol=2e-3, atol=1e-7)
    assert torch.allclose(getattr(ms[1], "xyz_encoding_1")[0].bias.grad.cpu(), golden["gr_full_f_xyz_encoding_1.0.bias"],
                          rtol=2e-3, atol=1e-7)


def test_training_step_loss_decreases_bf16(dev):
    """End-to-end: a few Adam steps through the bf16 HIP path reduce the loss on a fixed batch."""
    params = [O.make_params(5, 4.0, 0.2), O.make_params(6, 4.0, 0.2)]
    ms, emb = build_models(params, dev, "bf16")
    from nerf_pl_amd.models import render_rays
    rays = O.make_rays(3, 256, "blender").to(dev)
    tgt = torch.rand(256, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    opt = torch.optim.Adam([p for m in ms for p in m.parameters()], lr=2e-4)
    torch.manual_seed(0)
    losses = []
    for _ in range(8):
        res = render_rays(ms, emb, rays, 64, False, 1.0, 0.0, 64, 1024 * 32, True)
        loss = torch.nn.functional.mse_loss(res["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(res["rgb_fine"], tgt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0] - 1e-3, losses


def test_fused_adam_updates_are_seen(dev):
    """Regression: torch.optim.Adam(fused=True) (what NeRFSystem.configure_optimizers builds on the GPU) updates
    parameters WITHOUT bumping their version counters; the packed MFMA weight image must follow anyway."""
    from argparse import Namespace
    from nerf_pl_amd.system import NeRFSystem, fit
    hp = Namespace(N_samples=64, N_importance=64, use_disp=False, perturb=1.0, noise_std=0.0, chunk=1024 * 32, loss_type="mse",
                   lr=5e-4, weight_decay=0, decay_step=[100], decay_gamma=0.5, white_back=True)
    system = NeRFSystem(hp)
    system.nerf_coarse.load_state_dict(O.make_params(5, 4.0, 0.2))
    system.nerf_fine.load_state_dict(O.make_params(6, 4.0, 0.2))
    for m in system.models:
        m.mlp_dtype = "bf16"
    system = system.to(dev)
    rays = O.make_rays(3, 256, "blender").to(dev)
    tgt = torch.rand(256, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        before = system(rays)["rgb_fine"].clone()
    torch.manual_seed(0)
    losses = fit(system, [{"rays": rays, "rgbs": tgt}] * 12)
    assert isinstance(system.optimizer, torch.optim.Adam) and system.optimizer.defaults.get("fused")
    with torch.no_grad():
        after = system(rays)["rgb_fine"]
    assert (after - before).abs().max().item() > 1e-3, "render did not change after 12 fused-Adam steps"
    ls = [l.item() for l in losses]
    assert min(ls[-3:]) < ls[0], ls            # the optimizer is acting on the weights the kernels see
