"""GPU: NeRF shapes other than the reference's default (models/nerf.py:42-81 takes any D / W / skips / channel counts) run layer
by layer through csrc/linear.hip.  Checked against (1) a plain torch fp32 restatement of one layer, (2) vectors minted from the
real reference for three non-default configurations (tests/golden/reference_golden_arch.npz), (3) the fused kernels on the
default shape, (4) the CPU oracle on a deeper default-width network.  Tolerances: exact-fp32 MFMA 1e-4 class, bf16 3e-2 class."""
import os

import numpy as np
import pytest
import torch

from helpers import O, build_arch_models, build_models
from oracle.arch_cases import ARCHS, N_I, S_C

pytestmark = pytest.mark.gpu
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_golden_arch.npz")


@pytest.fixture(scope="module")
def garch():
    z = np.load(PATH, allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _act(name):
    return {"none": lambda v: v, "relu": torch.relu, "sigmoid": torch.sigmoid}[name]


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("n,segs,n_out,act", [(1, (5,), 1, "none"), (37, (63,), 80, "relu"), (1000, (63, 256), 256, "relu"),
                                               (513, (128, 27), 64, "relu"), (200, (64,), 3, "sigmoid"), (4097, (319,), 33, "none"),
                                               (64, (32, 32), 64, "sigmoid")])
def test_linear_act_vs_torch(dev, dtype, tol, n, segs, n_out, act):
    """One layer, forward and all three gradients, against torch fp32 on the same device (bf16: against torch on bf16-rounded
    operands for the forward, loose bound for the gradients).  Segments are column slices of a wider tensor (row stride != width)."""
    from nerf_pl_amd import ops
    g = torch.Generator().manual_seed(n * 7 + n_out)
    k_tot = sum(segs)
    wide = (torch.rand(n, k_tot + 9, generator=g) * 2 - 1).to(dev)
    w = ((torch.rand(n_out, k_tot, generator=g) * 2 - 1) / k_tot ** 0.5).to(dev).requires_grad_(True)
    b = ((torch.rand(n_out, generator=g) * 2 - 1) * 0.1).to(dev).requires_grad_(True)
    G = torch.randn(n, n_out, generator=g).to(dev)
    xs, col = [], 4
    for c in segs:
        xs.append(wide[:, col:col + c].detach().requires_grad_(True))
        col += c
    code = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "sigmoid": ops.ACT_SIGMOID}[act]
    y = ops.linear_act(xs, w, b, code, dtype)
    (y * G).sum().backward()
    got = [y.detach(), w.grad.clone(), b.grad.clone()] + [x.grad.clone() for x in xs]
    w.grad = b.grad = None
    xr = [x.detach().clone().requires_grad_(True) for x in xs]
    cat = torch.cat(xr, 1)
    if dtype == "bf16":
        yr = _act(act)(cat.bfloat16().float() @ w.bfloat16().float().t() + b)
    else:
        yr = _act(act)(cat.double() @ w.double().t() + b.double()).float()
    (yr * G).sum().backward()
    ref = [yr.detach(), w.grad, b.grad] + [x.grad for x in xr]
    names = ["y", "gw", "gb"] + ["gx%d" % i for i in range(len(xs))]
    for nm, a, r in zip(names, got, ref):
        assert a.shape == r.shape, nm
        scale = r.abs().max().item() + 1e-12
        bound = tol if (dtype == "fp32" or nm == "y") else 3e-2          # bf16 gradients: g and the operands are rounded too
        assert (a - r).abs().max().item() <= bound * scale + 1e-7, (nm, (a - r).abs().max().item(), scale)


def test_linear_act_empty_and_errors(dev):
    from nerf_pl_amd import ops
    w = torch.randn(8, 5, device=dev, requires_grad=True)
    b = torch.zeros(8, device=dev, requires_grad=True)
    y = ops.linear_act([torch.empty(0, 5, device=dev)], w, b, ops.ACT_RELU, "fp32")
    assert y.shape == (0, 8)
    y.sum().backward()
    assert w.grad.abs().max().item() == 0 and b.grad.abs().max().item() == 0
    with pytest.raises(ValueError):
        ops.linear_act([torch.empty(3, 4, device=dev)], w, b, ops.ACT_RELU, "fp32")
    with pytest.raises(ops.NerfHipError):
        ops.linear_act([torch.empty(3, 5)], w, b, ops.ACT_RELU, "fp32")


@pytest.mark.parametrize("tag", sorted(ARCHS))
def test_nerf_forward_and_grads_fp32_vs_reference_golden(garch, dev, tag):
    kw, seed = ARCHS[tag]
    arch = O.make_arch(**kw)
    (m,), _, _ = build_arch_models(arch, [seed], dev, "fp32")
    assert not m.is_default_arch()
    x = garch[tag + "/x"].to(dev).requires_grad_(True)
    out = m(x)
    ref = garch[tag + "/out"]
    assert out.shape == ref.shape
    assert torch.allclose(out.detach().cpu(), ref, rtol=1e-4, atol=1e-5), (out.detach().cpu() - ref).abs().max()
    (out * garch[tag + "/G"].to(dev)).sum().backward()
    gx = garch[tag + "/gx"]
    assert (x.grad.cpu() - gx).abs().max().item() <= 1e-4 * gx.abs().max().item() + 1e-8
    for n, p in m.named_parameters():
        r = garch[tag + "/g/" + n]
        assert (p.grad.cpu() - r).abs().max().item() <= 2e-4 * r.abs().max().item() + 1e-8, n
    with torch.no_grad():
        sig = m(x[:, :arch["in_xyz"]].detach(), sigma_only=True).cpu()
        lead = m(x.detach().view(4, -1, x.shape[1]))                       # leading dimensions, like nn.Linear
    assert torch.allclose(sig, garch[tag + "/sigma_only"], rtol=1e-4, atol=1e-5)
    assert lead.shape == (4, x.shape[0] // 4, 4) and torch.equal(lead.reshape(-1, 4), out.detach())


@pytest.mark.parametrize("tag", sorted(ARCHS))
def test_render_rays_and_training_grads_fp32_vs_reference_golden(garch, dev, tag):
    from nerf_pl_amd.models.rendering import render_rays
    kw, seed = ARCHS[tag]
    arch = O.make_arch(**kw)
    ms, embs, _ = build_arch_models(arch, [seed, seed + 1], dev, "fp32")
    rays = garch[tag + "/rays"].to(dev)
    res = render_rays(ms, embs, rays, S_C, False, 0, 0, N_I, 1000, True, False)            # chunk 1000: several MLP chunks
    assert sorted(res) == sorted(k[len(tag + "/render/"):] for k in garch if k.startswith(tag + "/render/"))
    for k, v in res.items():
        r = garch[tag + "/render/" + k]
        assert torch.allclose(v.detach().cpu(), r, rtol=1e-4, atol=1e-4), (k, (v.detach().cpu() - r).abs().max().item())
    tgt = garch[tag + "/target"].to(dev)
    loss = torch.nn.functional.mse_loss(res["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(res["rgb_fine"], tgt)
    loss.backward()
    assert abs(loss.item() - garch[tag + "/loss"].item()) <= 1e-4 * abs(garch[tag + "/loss"].item())
    for mi, m in enumerate(ms):
        for n, p in m.named_parameters():
            ref = garch[tag + "/rg%d/" % mi + n]
            dig = O.grad_digest(p.grad.cpu())
            scale = ref[1].abs().item() + 1e-12
            assert bool(((dig - ref).abs() <= 4e-3 * scale + 2e-3 * ref.abs() + 1e-9).all()), (mi, n, dig[:4], ref[:4])
    with torch.no_grad():
        tt = render_rays(ms, embs, rays, S_C, False, 0, 0, N_I, 1 << 15, True, True)
    assert sorted(tt) == sorted(k[len(tag + "/render_tt/"):] for k in garch if k.startswith(tag + "/render_tt/"))
    for k, v in tt.items():
        assert torch.allclose(v.cpu(), garch[tag + "/render_tt/" + k], rtol=1e-4, atol=1e-4), k


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-5), ("bf16", 3e-2)])
def test_layered_equals_fused_on_default_shape(dev, dtype, tol):
    """The layer-by-layer path evaluated on the DEFAULT shape agrees with the fused kernels (outputs and parameter gradients)."""
    from nerf_pl_amd.models.layered import nerf_forward
    p = O.make_params(91, 5.0, 0.2)
    (m,), _ = build_models([p], dev, dtype)
    x = (torch.rand(777, 90, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(dev)
    G = torch.randn(777, 4, generator=torch.Generator().manual_seed(4)).to(dev)
    a = m(x)
    (a * G).sum().backward()
    ga = {n: q.grad.clone() for n, q in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    b = nerf_forward(m, x)
    (b * G).sum().backward()
    assert (a - b).abs().max().item() <= tol * max(1.0, a.abs().max().item())
    for n, q in m.named_parameters():
        r = ga[n]
        rel = (q.grad - r).norm().item() / (r.norm().item() + 1e-20)
        assert rel <= (1e-4 if dtype == "fp32" else 3e-2), (n, rel)
    with torch.no_grad():
        assert (m(x[:, :63], sigma_only=True) - nerf_forward(m, x[:, :63], True)).abs().max().item() <= tol * max(1.0, a[:, 3].abs().max().item())


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 3e-2), ("bf16_f8", 3e-2)])
def test_deeper_default_width_vs_oracle(dev, dtype, tol):
    """D = 9, W = 256, skips = [3, 6] with the default embeddings: render_rays (perturb = 1, noise_std = 1, replayed draws)
    against the CPU oracle at 256 rays x (32 + 32) samples."""
    from helpers import hip_render
    arch = O.make_arch(D=9, W=256, skips=(3, 6))
    ms, embs, params = build_arch_models(arch, [61, 62], dev, dtype, 4.0, 0.2)
    B, S, N = 256, 32, 32
    rays = O.make_rays(5, B, "ndc")
    rng = O.draw_rng(9, B, S, N, 1.0)
    kw = dict(N_samples=S, use_disp=False, perturb=1.0, noise_std=1.0, N_importance=N, white_back=False, test_time=False)
    ref = O.render_rays(params, rays, S, False, 1.0, 1.0, N, False, False, rng=rng, arch=arch)
    with torch.no_grad():
        res = hip_render(ms, embs, rays, kw, rng, dev)
    assert sorted(res) == sorted(ref)
    for k in ref:
        got = res[k].cpu()
        bad = ~torch.isclose(got, ref[k], rtol=tol, atol=tol)
        assert bad.float().mean().item() <= (4e-3 if dtype == "fp32" else 2e-2), (k, bad.float().mean().item(), (got - ref[k]).abs().max().item())


def test_training_with_a_non_default_shape_converges(dev):
    """NeRFSystem-style loop on a small network (D = 4, W = 128, 6/2 bands): the modular step (render_rays -> MSELoss -> backward ->
    FlatAdam) on a fixed batch reduces the loss; the fused step refuses the shape and is not used."""
    from nerf_pl_amd.losses import MSELoss
    from nerf_pl_amd.models import train_step
    from nerf_pl_amd.models.rendering import render_rays
    from nerf_pl_amd.optim import FlatAdam
    arch = O.make_arch(D=4, W=128, N_freq_xyz=6, N_freq_dir=2, skips=(2,))
    ms, embs, _ = build_arch_models(arch, [71, 72], dev, "bf16", 1.0, 0.0)
    assert not train_step.fusable(ms, embs, MSELoss())
    opt = FlatAdam(ms, lr=1e-3)
    rays = O.make_rays(3, 512, "blender").to(dev)
    tgt = torch.rand(512, 3, generator=torch.Generator().manual_seed(1)).to(dev) * 0.5 + 0.25
    torch.manual_seed(0)
    losses = []
    for _ in range(40):
        opt.zero_grad(set_to_none=True)
        res = render_rays(ms, embs, rays, 32, False, 1.0, 1.0, 32, 1 << 15, True, False)
        loss = MSELoss()(res, tgt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.85 * losses[0], (losses[0], losses[-1])          # measured 0.72 (random per-ray target colours)


def test_sigma_grid_non_default_vs_oracle(dev):
    from nerf_pl_amd.grid import sigma_grid
    arch = O.make_arch(D=3, W=64, N_freq_xyz=4, N_freq_dir=1, skips=(2,))
    (m,), embs, (p,) = build_arch_models(arch, [81], dev, "fp32", 3.0, 0.1)
    N = 12
    got = sigma_grid(m, N, (-1, 1), (-1.2, 1.2), (-0.8, 0.8), clamp=True, embedding_xyz=embs[0]).cpu()
    xs, ys, zs = (torch.from_numpy(np.linspace(lo, hi, N).astype(np.float32)) for lo, hi in ((-1, 1), (-1.2, 1.2), (-0.8, 0.8)))
    pts = torch.from_numpy(np.stack(np.meshgrid(xs.numpy(), ys.numpy(), zs.numpy()), -1).reshape(-1, 3))   # extract_color_mesh.py:118-122
    ref = O.mlp_forward(p, O.posenc(pts, 4), sigma_only=True, arch=arch).clamp(min=0).view(N, N, N)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5), (got - ref).abs().max()
    with pytest.raises(ValueError):
        sigma_grid(m, N, (-1, 1), (-1, 1), (-1, 1))


def test_nerfsystem_with_a_non_default_shape_graphed_equals_eager(dev):
    """NeRFSystem whose models / embeddings were swapped for a non-default configuration (what a subclass of the reference's
    system would do): training_step falls back to the modular graph (render_rays layer by layer -> MSELoss), FlatAdam updates
    the 2 x (2 D + 8) tensors, and the whole step replays as a hipGraph with the same losses and weights as eager issue."""
    from argparse import Namespace
    from nerf_pl_amd.models import Embedding, NeRF
    from nerf_pl_amd.system import GraphedTrainStep, NeRFSystem
    arch = O.make_arch(D=3, W=64, N_freq_xyz=5, N_freq_dir=2, skips=(1,))
    gen = torch.Generator().manual_seed(2)
    batches = [{"rays": O.make_rays(20 + i, 96, "blender").to(dev), "rgbs": torch.rand(96, 3, generator=gen).to(dev)} for i in range(7)]
    finals, losses = [], []
    for graphed in (False, True):
        hp = Namespace(N_samples=16, N_importance=16, use_disp=False, perturb=0.0, noise_std=0.0, chunk=1024, loss_type="mse",
                       lr=1e-3, weight_decay=0, decay_step=[100], decay_gamma=0.5, white_back=True)
        system = NeRFSystem(hp)
        for name, seed in (("nerf_coarse", 31), ("nerf_fine", 32)):
            m = NeRF(D=arch["D"], W=arch["W"], in_channels_xyz=arch["in_xyz"], in_channels_dir=arch["in_dir"], skips=list(arch["skips"]))
            m.load_state_dict(O.make_params(seed, 4.0, 0.2, arch=arch))
            m.mlp_dtype = "fp32"
            setattr(system, name, m)
        system.models = [system.nerf_coarse, system.nerf_fine]
        system.embedding_xyz, system.embedding_dir = Embedding(3, 5), Embedding(3, 2)
        system.embeddings = [system.embedding_xyz, system.embedding_dir]
        system = system.to(dev)
        assert not system._fused_step_ok(batches[0]["rays"])
        (opt,), _ = system.configure_optimizers()
        stepper = GraphedTrainStep(system, opt, warmup=2 if graphed else 10 ** 9)
        ls = [stepper(b)["loss"].item() for b in batches]
        assert (stepper.graph is not None) == graphed
        losses.append(ls)
        finals.append({k: v.detach().cpu().clone() for k, v in system.state_dict().items()})
    assert losses[0] == pytest.approx(losses[1], rel=1e-5), (losses[0], losses[1])
    assert losses[0][-1] < losses[0][0]
    for k in finals[0]:
        assert torch.allclose(finals[0][k], finals[1][k], rtol=1e-5, atol=1e-7), k
