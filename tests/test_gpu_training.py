"""GPU: backward kernels (mlp_bwd_chain + mlp_bwd_dw + reduce, composite_bwd) against autograd through
the CPU oracle and against the gradient digests minted from the real reference.

Tolerances: fp32 MFMA path 2e-4 of each tensor's max |grad| (different fp32 summation order over up to
~10^4 points); reduced-precision paths: per-tensor gradient cosine >= 0.99 (bf16) / 0.98 (8-bit dW operands) — they are the
roofline configurations, licensed by PSNR@step (tests/test_gpu_psnr_gate.py) rather than by gradient bits."""
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


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-4), ("bf16", 4e-2), ("bf16_f8", 4e-2)])
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
    worst = 1.0
    for name, prm in m.named_parameters():
        assert prm.grad is not None, name
        r = ref[name]
        gq = prm.grad.cpu()
        if dtype == "fp32":
            err = (gq - r).abs().max().item()
            assert err <= tol * r.abs().max().item() + 1e-7, (dtype, n, name, err, r.abs().max().item())
        else:
            # reduced precision: the DIRECTION of every gradient tensor (the bound of tests/test_gpu_bf16.py, here also at the
            # ragged sizes n = 1, 33, 300): cosine >= 0.99 for bf16, >= 0.98 with the 8-bit dW operands (>= 0.98 for both at n = 1); the relative L2 error
            # (0.1-11 % measured, deepest layers worst: the bf16 FORWARD activations, hence ReLU gates, already differ from the
            # fp32 oracle's) is printed, not gated
            rel = (gq - r).norm().item() / (r.norm().item() + 1e-12)
            cos = torch.nn.functional.cosine_similarity(gq.flatten(), r.flatten(), dim=0).item()
            worst = min(worst, cos)
            # (n = 1: a single point, nothing averages — measured 0.987 / 0.985; n = 33: 0.991 / 0.983)
            assert cos >= ((0.99 if dtype == "bf16" else 0.98) if n >= 33 else 0.98), (dtype, n, name, rel, cos)
    if dtype != "fp32":
        print("worst per-tensor gradient cosine, %s n=%d: %.4f" % (dtype, n, worst))


def test_f8_storage_gradients_track_bf16(dev):
    """The fp8-storage mode changes only the operands of the weight-gradient GEMM (e4m3, one power-of-two scale per 32 points
    x 32 features): forward output identical to bf16, every gradient tensor within a few % (relative L2) of the bf16 one."""
    n = 3000
    g = torch.Generator().manual_seed(7)
    p = O.make_params(21, 3.0, 0.1)
    pts = torch.rand(n, 3, generator=g) * 4 - 2
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    x = torch.cat([O.posenc(pts, 10), O.posenc(dirs, 4)], 1).to(dev)
    g_out = torch.randn(n, 4, generator=g).to(dev)
    grads, outs = {}, {}
    for dt in ("bf16", "bf16_f8"):
        (m,), _ = build_models([p], dev, dt)
        out = m(x)
        (out * g_out).sum().backward()
        outs[dt] = out.detach()
        grads[dt] = {k: v.grad.clone() for k, v in m.named_parameters()}
    assert torch.equal(outs["bf16"], outs["bf16_f8"])
    worst = 0.0
    for k in grads["bf16"]:
        a, b = grads["bf16_f8"][k], grads["bf16"][k]
        rel = (a - b).norm().item() / (b.norm().item() + 1e-20)
        worst = max(worst, rel)
        assert rel <= 0.15, (k, rel)        # measured: <= 0.095 with this test's random-sign g_out (heavy cancellation in the sums);
                                            # e4m3 keeps 3 mantissa bits and the rounding errors average over the points
    print("fp8-storage vs bf16 gradients: worst relative L2 difference %.4f" % worst)


@pytest.mark.parametrize("prefix", ["gr", "gr3", "gr4"])
def test_training_grads_fp32_vs_reference_golden(golden, dev, prefix):
    """Gradients of the training loss w.r.t. all 48 parameter tensors against digests minted from the real reference:
    gr = 64+64 Blender noise_std=1; gr3 = BASELINE configs[2] shape (64+128, perturb=1, noise_std=0, white background);
    gr4 = configs[3] shape (NDC rays, noise_std=1, black background).  fp32 path, 2e-4-class tolerances."""
    params, rays, kw, rng = case_from_golden(golden, None, prefix=prefix)
    ms, emb = build_models(params, dev, "fp32")
    res = hip_render(ms, emb, rays, kw, rng, dev)
    tgt = golden[f"{prefix}_target"].to(dev)
    loss = torch.nn.functional.mse_loss(res["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(res["rgb_fine"], tgt)
    loss.backward()
    _check_against_reference_gradients(golden, prefix, ms, loss, res["rgb_fine"])


@pytest.mark.parametrize("prefix", ["gr", "gr3", "gr4"])
def test_fused_training_node_fp32_vs_reference_golden(golden, dev, prefix):
    """The node bench.py TIMES — models/train_step.render_rays_train: one autograd node, the fused launches of round 4 (coarse
    depths in the MLP prologue, compositing + loss gradient + compositing backward + fine depths / loss per pass, one dW and one
    reduce launch for both models) — pinned DIRECTLY to the reference-minted gradients of the same three cases, on the draws the
    reference consumed when they were minted (round 3 reached the goldens only through the modular graph)."""
    from helpers import fused_draws
    from nerf_pl_amd.models.train_step import render_rays_train
    params, rays, kw, rng = case_from_golden(golden, None, prefix=prefix)
    ms, emb = build_models(params, dev, "fp32")
    tgt = golden[f"{prefix}_target"].to(dev)
    res, loss, out3 = render_rays_train(ms, emb, rays.to(dev), tgt, kw["N_samples"], kw["use_disp"], kw["perturb"], kw["noise_std"],
                                        kw["N_importance"], kw["white_back"], draws=fused_draws(rng, kw, dev))
    loss.backward()
    assert out3[0].item() == loss.item()
    _check_against_reference_gradients(golden, prefix, ms, loss, res["rgb_fine"])


# tensors of the FINE model upstream of its density head (layers 1-8, sigma): see the docstring below
def _fine_trunk(name):
    return name.startswith("f.") and not name.startswith(("f.rgb", "f.dir", "f.xyz_encoding_final"))


@pytest.mark.parametrize("form", ["modular", "fused"])
def test_training_grads_fp32_all_48_tensors_in_full(golden, golden_grads, dev, form):
    """configs[2] shape (gr3: 64 + 128 samples, perturb = 1, noise_std = 0, white background): EVERY one of the 48 gradient tensors
    of the training loss, element for element, against the tensors the reference's own autograd produced
    (tests/golden/reference_golden_grads.npz; 1,191,688 values) — the modular render_rays graph and the fused training node that
    bench.py times.
    * the coarse model's 24 tensors and the fine model's colour branch (xyz_encoding_final, dir_encoding, rgb): 2e-4 of the
      tensor's max |g| (+ 5e-9).  Measured: <= 2e-6 (coarse), <= 5.5e-5 (fine colour branch) — the fp32 MFMA path sums the
      points in another order than ATen's GEMMs, nothing else differs.
    * the fine model's trunk and density head (layers 1-8, sigma): 2e-2 of max |g|, measured <= 8.8e-3.  These are not looser
      kernels, they are a worse-conditioned QUESTION: the fine pass's depths cluster (deltas down to 1e-6), its density
      gradient is d alpha / d sigma = delta e^(-delta sigma), and delta = z[i+1] - z[i] moves by 0.1-50 % when a depth moves by
      one ulp.  The depths of the two implementations differ by ulps because the coarse weights do (last bits of an fp32
      GEMM).  Quantified on the CPU (tests/test_oracle_golden.py::test_fine_pass_conditioning): nudging 30 % of the fine
      depths of this very case by +-1 ulp moves these tensors of the REFERENCE's arithmetic by up to 1.1e-3 of their maximum,
      and fp32 against fp64 on IDENTICAL depths by 4.6e-4 (coarse model: 6e-7).  The next test removes the depths from the
      comparison and holds these tensors to 2e-3."""
    rows = _gr3_rows(golden, golden_grads, dev, form)
    print("gr3, %s step, fp32: max |g - g_ref| / max |g_ref| per tensor:" % form)
    for row in sorted(rows, key=lambda t: -t[1]):
        print("   %-28s %.2e  (err %.2e, max|g| %.2e, %d elements)" % row)
    for name, ratio, err, mx, _ in rows:
        assert err <= (2e-2 if _fine_trunk(name) else 2e-4) * mx + 5e-9, (form, name, err, mx)


def test_fine_pass_grads_fp32_on_identical_depths(golden, dev):
    """The fine model's 24 gradient tensors with the depths taken OUT of the comparison: the HIP coarse pass produces z_fine,
    the HIP fine pass (fused MLP + compositing, autograd through both) and the CPU oracle's fine pass (rendering.py:231-242 on
    the same rays) both evaluate THOSE depths.  What is left is fp32 arithmetic on identical inputs: every tensor within 2e-3
    of its max |g| (fp32 against fp64 on identical depths is 4.6e-4 for the reference's own arithmetic on this case)."""
    from nerf_pl_amd import ops
    from nerf_pl_amd.models.mlp_autograd import mlp_rays
    params, rays, kw, rng = case_from_golden(golden, None, prefix="gr3")
    ms, emb = build_models(params, dev, "fp32")
    tgt = golden["gr3_target"]
    rd = rays.to(dev)
    with torch.no_grad():
        z = ops.sample_coarse_z(rd, kw["N_samples"], False, 1.0, rng["perturb_rand"].to(dev))
        raw_c = mlp_rays(ms[0], rd, z, False)
        w_c = ops.composite(raw_c, z, rd, None, 0.0, True)[0]
        zf = ops.fine_z(z, w_c, kw["N_importance"], u=rng["u"].to(dev))
    raw_f = mlp_rays(ms[1], rd, zf, False)
    rgb_f = ops.composite(raw_f, zf, rd, None, 0.0, True)[2]
    torch.nn.functional.mse_loss(rgb_f, tgt.to(dev)).backward()
    pf = {k: v.clone().requires_grad_(True) for k, v in params[1].items()}
    f = O._infer(pf, rays, zf.cpu(), O.posenc(rays[:, 3:6], 4), None, True, False)
    torch.mean((f["rgb"] - tgt) ** 2).backward()
    assert (rgb_f.detach().cpu() - f["rgb"].detach()).abs().max().item() <= 1e-5
    worst = ("", 0.0)
    for n, prm in ms[1].named_parameters():
        ref = pf[n].grad
        ratio = (prm.grad.cpu() - ref).abs().max().item() / ref.abs().max().item()
        worst = max(worst, (n, ratio), key=lambda t: t[1])
        assert ratio <= 2e-3, (n, ratio)
    print("fine pass on identical depths, fp32: worst max |g - g_oracle| / max |g_oracle| = %.2e (%s)" % (worst[1], worst[0]))


def _gr3_rows(golden, golden_grads, dev, form):
    from helpers import fused_draws
    from nerf_pl_amd.models.train_step import render_rays_train
    params, rays, kw, rng = case_from_golden(golden, None, prefix="gr3")
    ms, emb = build_models(params, dev, "fp32")
    tgt = golden["gr3_target"].to(dev)
    if form == "fused":
        _, loss, _ = render_rays_train(ms, emb, rays.to(dev), tgt, kw["N_samples"], kw["use_disp"], kw["perturb"], kw["noise_std"],
                                       kw["N_importance"], kw["white_back"], draws=fused_draws(rng, kw, dev))
    else:
        res = hip_render(ms, emb, rays, kw, rng, dev)
        loss = torch.nn.functional.mse_loss(res["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(res["rgb_fine"], tgt)
    loss.backward()
    rows, n_el = [], 0
    for tag, m in (("c", ms[0]), ("f", ms[1])):
        for n, prm in m.named_parameters():
            ref = golden_grads[f"gr3_grad_{tag}_{n}"]
            err = (prm.grad.cpu() - ref).abs().max().item()
            rows.append((tag + "." + n, err / (ref.abs().max().item() + 1e-30), err, ref.abs().max().item(), ref.numel()))
            n_el += ref.numel()
    assert n_el == 1191688
    return rows


def _check_against_reference_gradients(golden, prefix, ms, loss, rgb_fine):
    assert abs(loss.item() - golden[f"{prefix}_loss"].item()) <= 1e-4 * abs(golden[f"{prefix}_loss"].item())
    assert torch.allclose(rgb_fine.detach().cpu(), golden[f"{prefix}_rgb_fine"], rtol=1e-4, atol=1e-4)
    for tag, m in (("c", ms[0]), ("f", ms[1])):
        for n, prm in m.named_parameters():
            dig = O.grad_digest(prm.grad.cpu())
            ref = golden[f"{prefix}_{tag}_{n}"]
            scale = ref[1].abs().item() + 1e-12          # l2 norm of the reference gradient tensor
            # digest = [sum, l2, 8 head, 8 tail]; the signed sum of up to 81k elements cancels heavily,
            # so it is compared against the tensor's l2 norm
            # (|sum| can exceed l2 by up to sqrt(numel), so each entry also gets its own 2e-3 rel band)
            assert bool(((dig - ref).abs() <= 4e-3 * scale + 2e-3 * ref.abs() + 1e-9).all()), (prefix, tag, n, dig[:4], ref[:4])
            # l2 norms agree to 3e-3 (measured <= 1.1e-3: the tiny sigma.weight gradient, 7e-6, is a heavily cancelling sum
            # whose fp32 rounding depends on the split-K tree)
            assert abs(dig[1].item() - ref[1].item()) <= 3e-3 * scale, (prefix, tag, n, dig[1], ref[1])
    full = [("c_sigma.weight", ms[0].sigma.weight), ("f_rgb.0.weight", ms[1].rgb[0].weight),
            ("f_xyz_encoding_1.0.bias", getattr(ms[1], "xyz_encoding_1")[0].bias)]
    if prefix != "gr":
        full.append(("f_dir_encoding.0.weight", ms[1].dir_encoding[0].weight))
    for name, prm in full:
        ref = golden[f"{prefix}_full_{name}"]
        err = (prm.grad.cpu() - ref).abs().max().item()
        # 2e-4 of the tensor's max |grad|, with an absolute floor of 5e-9: the smallest tensors (first-layer bias, max 3e-7)
        # are cancelling sums of ~6000 terms of 1e-6, whose fp32 accumulation error alone is ~2e-9 in either implementation
        assert err <= 2e-4 * ref.abs().max().item() + 5e-9, (prefix, name, err, ref.abs().max().item())


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
                   lr=5e-4, weight_decay=0, decay_step=[100], decay_gamma=0.5, white_back=True, flat_optimizer=False)
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


def test_fused_mse_psnr_matches_reference_formulas(dev):
    """nerfhip_mse_psnr == losses.py:9-14 + metrics.py:4-13 (+ autograd of the MSE), with and without a fine image."""
    from nerf_pl_amd import ops
    g = torch.Generator().manual_seed(4)
    for B in (1, 7, 1024, 5000):
        c = torch.rand(B, 3, generator=g)
        f = torch.rand(B, 3, generator=g)
        t = torch.rand(B, 3, generator=g)
        for fine in (f, None):
            cc = c.clone().requires_grad_(True)
            ff = fine.clone().requires_grad_(True) if fine is not None else None
            ref = torch.nn.functional.mse_loss(cc, t) + (torch.nn.functional.mse_loss(ff, t) if ff is not None else 0)
            ref.backward()
            ref_psnr = -10 * torch.log10(torch.mean(((ff if ff is not None else cc).detach() - t) ** 2))
            cd = c.to(dev).requires_grad_(True)
            fd = fine.to(dev).requires_grad_(True) if fine is not None else None
            loss, out3 = ops.mse_psnr(cd, fd, t.to(dev))
            (loss * 3.0).backward()                              # upstream factor must be applied
            assert abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item())
            assert abs(out3[1].item() - ref_psnr.item()) <= 1e-4
            assert torch.allclose(cd.grad.cpu(), 3.0 * cc.grad, rtol=1e-6, atol=1e-9)
            if fine is not None:
                assert torch.allclose(fd.grad.cpu(), 3.0 * ff.grad, rtol=1e-6, atol=1e-9)


def test_flat_adam_equals_per_tensor_adam(dev):
    """FlatAdam (one flat tensor per model, grads adopted from the dW-reduce buffer, HIP update kernel) == torch Adam over
    the 48 tensors.  ONE step from identical weights and gradients (perturb=0: deterministic kernels), so the comparison
    is of the two update kernels; over several steps the trajectories themselves separate (entries whose gradient is
    ~0 flip sign under 1e-8 perturbations and Adam moves them by +-lr), which is not an optimizer difference."""
    from argparse import Namespace
    from nerf_pl_amd.system import NeRFSystem, fit
    rays = O.make_rays(3, 192, "blender").to(dev)
    tgt = torch.rand(192, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    finals = []
    for flat in (True, False):
        hp = Namespace(N_samples=64, N_importance=64, use_disp=False, perturb=0.0, noise_std=0.0, chunk=1024 * 32,
                       loss_type="mse", lr=5e-4, weight_decay=0, decay_step=[100], decay_gamma=0.5, white_back=True,
                       flat_optimizer=flat)
        system = NeRFSystem(hp)
        system.nerf_coarse.load_state_dict(O.make_params(5, 4.0, 0.2))
        system.nerf_fine.load_state_dict(O.make_params(6, 4.0, 0.2))
        for m in system.models:
            m.mlp_dtype = "fp32"
        system = system.to(dev)
        torch.manual_seed(0)
        losses = fit(system, [{"rays": rays, "rgbs": tgt}] * 1)
        assert type(system.optimizer).__name__ == ("FlatAdam" if flat else "Adam")
        finals.append({k: v.detach().cpu().clone() for k, v in system.state_dict().items()})
        more = fit(system, [{"rays": rays, "rgbs": tgt}] * 5)         # and it keeps training
        assert more[-1].item() < losses[0].item()
    assert finals[0].keys() == finals[1].keys()
    worst = max((finals[0][k] - finals[1][k]).abs().max().item() for k in finals[0])
    print("FlatAdam vs torch fused Adam, one step: max |param diff| %.3e (the update is 5e-4)" % worst)
    for k in finals[0]:
        assert torch.allclose(finals[0][k], finals[1][k], rtol=1e-6, atol=2e-7), k
    # the flat path really moved the weights
    assert not torch.equal(finals[0]["nerf_fine.sigma.weight"], O.make_params(6, 4.0, 0.2)["sigma.weight"])


def test_adam_kernel_vs_float64_adam(dev):
    """nerfhip_adam_step against torch.optim.Adam run in float64 on the CPU (the arithmetic of utils/__init__.py:18-20),
    6 steps, weight decay on, gradients spanning 1e-9 .. 1 (the eps = 1e-8 regime included)."""
    from nerf_pl_amd.models import NeRF
    from nerf_pl_amd.optim import FlatAdam
    m = NeRF()
    m.load_state_dict(O.make_params(5))
    m = m.to(dev)
    ref = [p.detach().cpu().double().clone().requires_grad_(True) for p in m.parameters()]
    opt = FlatAdam([m], lr=5e-4, eps=1e-8, weight_decay=1e-4)
    ropt = torch.optim.Adam(ref, lr=5e-4, eps=1e-8, weight_decay=1e-4)
    g = torch.Generator().manual_seed(0)
    for step in range(6):
        for p, r in zip(m.parameters(), ref):
            gr = torch.randn(p.shape, generator=g) * (10.0 ** torch.randint(-9, 1, p.shape, generator=g).float())
            p.grad = gr.to(dev)
            r.grad = gr.double()
        opt.step()
        ropt.step()
    worst = 0.0
    for p, r in zip(m.parameters(), ref):
        worst = max(worst, (p.detach().cpu().double() - r.detach()).abs().max().item())
    print("adam kernel vs float64 Adam after 6 steps: max |param diff| %.3e (one update is 5e-4)" % worst)
    assert worst <= 2e-7, worst                             # fp32 rounding of params ~0.1 is 7e-9 per step
    assert float(opt.dev_state[0]) == 6.0


def test_flat_adam_state_dict_roundtrips_with_torch_adam(dev):
    """FlatAdam speaks torch.optim.Adam's per-parameter state_dict (48 entries in model.parameters() order): state
    saved here loads into the reference's optimizer and continues identically, and the other way round."""
    from nerf_pl_amd.models import NeRF
    from nerf_pl_amd.optim import FlatAdam

    def grads_for(models, seed):
        g = torch.Generator().manual_seed(seed)
        for m in models:
            for p in m.parameters():
                p.grad = (torch.randn(p.shape, generator=g) * 1e-2).to(dev)

    def fresh():
        ms = []
        for s_ in (5, 6):
            m = NeRF()
            m.load_state_dict(O.make_params(s_, 4.0, 0.2))
            ms.append(m.to(dev))
        return ms

    # A: 3 FlatAdam steps -> state_dict -> torch Adam, 2 more steps.   B: 5 torch Adam steps.
    ma, mb = fresh(), fresh()
    fa = FlatAdam(ma, lr=5e-4, weight_decay=1e-3)
    tb = torch.optim.Adam([p for m in mb for p in m.parameters()], lr=5e-4, eps=1e-8, weight_decay=1e-3)
    for i in range(3):
        grads_for(ma, i); fa.step()
        grads_for(mb, i); tb.step()
    sd = fa.state_dict()
    assert len(sd["state"]) == 48 and sd["param_groups"][0]["params"] == list(range(48))
    assert all(float(st["step"]) == 3.0 for st in sd["state"].values())
    assert tuple(sd["state"][1]["exp_avg"].shape) == (256,)               # parameters() order: weight, bias, weight, ...
    ta = torch.optim.Adam([p for m in ma for p in m.parameters()], lr=5e-4, eps=1e-8, weight_decay=1e-3)
    ta.load_state_dict(sd)
    for i in range(3, 5):
        grads_for(ma, i); ta.step()
        grads_for(mb, i); tb.step()
    for a, b in zip([p for m in ma for p in m.parameters()], [p for m in mb for p in m.parameters()]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    # the other way: torch Adam state -> FlatAdam, continue, compare with torch Adam continuing
    mc = fresh()
    fc = FlatAdam(mc, lr=5e-4, weight_decay=1e-3)
    for m_src, m_dst in zip(mb, mc):
        with torch.no_grad():
            for ps, pd in zip(m_src.parameters(), m_dst.parameters()):
                pd.copy_(ps)
    fc.load_state_dict(tb.state_dict())
    for i in range(5, 7):
        grads_for(mc, i); fc.step()
        grads_for(mb, i); tb.step()
    for a, b in zip([p for m in mc for p in m.parameters()], [p for m in mb for p in m.parameters()]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    assert float(fc.dev_state[0]) == 7.0


def test_flat_adam_realiases_detached_parameters(dev):
    """A later `p.data = ...` (what model.float()/.to() do) breaks the aliasing with the flat buffer; step() must notice
    and re-alias instead of silently updating storage the kernels no longer read."""
    from nerf_pl_amd.models import NeRF
    from nerf_pl_amd.optim import FlatAdam
    m = NeRF()
    m.load_state_dict(O.make_params(5))
    m = m.to(dev)
    opt = FlatAdam([m], lr=1e-2)
    w = m.sigma.weight
    w.data = w.data.clone() * 2.0                          # detached storage with new values
    before = w.detach().clone()
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    off = sum(p.numel() for p in m.flat_params()[:10])
    assert w.data_ptr() == opt.flats[0].data_ptr() + 4 * off          # aliased again
    assert torch.allclose(w.detach(), before - 1e-2, rtol=0, atol=1e-6)   # first Adam step = -lr * sign(g), from the NEW values


def test_sigma_only_forward_is_differentiable(dev):
    """NeRF.forward(x, sigma_only=True) and render_rays(test_time=True) under grad: the density branch gets gradients
    (the reference's sigma-only forward is an ordinary differentiable module call, nerf.py:103-114)."""
    p = O.make_params(21, 3.0, 0.1)
    (m,), emb = build_models([p], dev, "fp32")
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(200, 3, generator=g) * 4 - 2
    x = O.posenc(pts, 10)
    gs = torch.randn(200, 1, generator=g)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    (O.mlp_forward(pr, x, sigma_only=True) * gs).sum().backward()
    out = m(x.to(dev), sigma_only=True)
    assert out.shape == (200, 1) and out.requires_grad
    (out * gs.to(dev)).sum().backward()
    for name, prm in m.named_parameters():
        ref = pr[name].grad
        if ref is None:                                     # colour branch: untouched by the density output
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
        else:
            assert (prm.grad.cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-7, name


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gradient_wrt_preembedded_inputs(dev, dtype):
    """NeRF.forward is differentiable w.r.t. its (pre-embedded) input like the reference module (nerf.py:100-124):
    nerfhip_mlp_dx_embedded against autograd through the oracle; in the fp8-storage mode such a call saves in bf16."""
    n = 700
    g = torch.Generator().manual_seed(5)
    p = O.make_params(21, 3.0, 0.1)
    pts = torch.rand(n, 3, generator=g) * 4 - 2
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    x = torch.cat([O.posenc(pts, 10), O.posenc(dirs, 4)], 1)
    g_out = torch.randn(n, 4, generator=g)
    x0 = x.clone().requires_grad_(True)
    (O.mlp_forward(p, x0) * g_out).sum().backward()
    (m,), _ = build_models([p], dev, dtype)
    x1 = x.clone().to(dev).requires_grad_(True)
    (m(x1) * g_out.to(dev)).sum().backward()
    assert x1.grad is not None and x1.grad.shape == (n, 90)
    ref, got = x0.grad, x1.grad.cpu()
    if dtype == "fp32":
        assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item(), (got - ref).abs().max().item()
    else:
        cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
        assert cos >= 0.99, cos
    assert all(q.grad is not None for q in m.parameters())           # parameter gradients still flow
    # sigma-only forward: d sigma / d x_xyz
    x2 = x[:, :63].clone().requires_grad_(True)
    (O.mlp_forward(p, x2, sigma_only=True) * g_out[:, 3:]).sum().backward()
    x3 = x[:, :63].clone().to(dev).requires_grad_(True)
    (m(x3, sigma_only=True) * g_out[:, 3:].to(dev)).sum().backward()
    if dtype == "fp32":
        assert (x3.grad.cpu() - x2.grad).abs().max().item() <= 2e-4 * x2.grad.abs().max().item()
    # the 8-bit storage mode: a call whose input requires grad saves its tensors in bf16 (the 8-bit mode keeps dY only as e5m2
    # copies for the dW GEMM), so d/dx and the parameter gradients of THAT call equal the bf16 mode's bit for bit
    if dtype == "bf16":
        res = {}
        for dt in ("bf16", "bf16_f8"):
            (mm,), _ = build_models([p], dev, dt)
            x4 = x.clone().to(dev).requires_grad_(True)
            (mm(x4) * g_out.to(dev)).sum().backward()
            res[dt] = (x4.grad.clone(), [q.grad.clone() for q in mm.parameters()])
        assert torch.equal(res["bf16"][0], res["bf16_f8"][0])
        assert all(torch.equal(a, b) for a, b in zip(res["bf16"][1], res["bf16_f8"][1]))


def test_empty_batch_gradients_are_zero(dev):
    """n == 0 launches nothing: the 24 gradients of an empty batch must be zeros, not uninitialised memory."""
    (m,), _ = build_models([O.make_params(5)], dev, "bf16")
    out = m(torch.zeros(0, 90, device=dev))
    assert out.shape == (0, 4)
    out.sum().backward()
    for name, prm in m.named_parameters():
        assert prm.grad is not None and float(prm.grad.abs().sum()) == 0.0, name


def test_graphed_train_step_equals_eager(dev):
    """hipGraph replay of the whole training step (GraphedTrainStep) == eager issue, step for step
    (perturb=0: no RNG in the math; same kernels, same inputs => same weights)."""
    from argparse import Namespace
    from nerf_pl_amd.system import GraphedTrainStep, NeRFSystem
    gen = torch.Generator().manual_seed(1)
    batches = [{"rays": O.make_rays(10 + i, 128, "blender").to(dev), "rgbs": torch.rand(128, 3, generator=gen).to(dev)}
               for i in range(8)]
    finals, losses = [], []
    for graphed in (False, True):
        hp = Namespace(N_samples=64, N_importance=64, use_disp=False, perturb=0.0, noise_std=0.0, chunk=1024 * 32,
                       loss_type="mse", lr=5e-4, weight_decay=0, decay_step=[100], decay_gamma=0.5, white_back=True)
        system = NeRFSystem(hp)
        system.nerf_coarse.load_state_dict(O.make_params(5, 4.0, 0.2))
        system.nerf_fine.load_state_dict(O.make_params(6, 4.0, 0.2))
        for m in system.models:
            m.mlp_dtype = "bf16"
        system = system.to(dev)
        (opt,), _ = system.configure_optimizers()
        stepper = GraphedTrainStep(system, opt, warmup=2 if graphed else 10 ** 9)
        ls = []
        for b in batches:
            ls.append(stepper(b)["loss"].item())
        assert (stepper.graph is not None) == graphed
        losses.append(ls)
        finals.append({k: v.detach().cpu().clone() for k, v in system.state_dict().items()})
    assert losses[0] == pytest.approx(losses[1], rel=1e-5), (losses[0], losses[1])
    for k in finals[0]:
        assert torch.allclose(finals[0][k], finals[1][k], rtol=1e-5, atol=1e-7), k


def test_graphed_step_with_rccl_allreduce_world1(dev):
    """The N>1 training step (gradient all-reduce on the flat buffers through backend nccl = RCCL) inside the hipGraph
    capture, exercised on one GPU with a world_size-1 process group (8-GPU runs belong to the driver)."""
    import os
    import socket
    import torch.distributed as dist
    from argparse import Namespace
    from nerf_pl_amd.parallel import GradSync
    from nerf_pl_amd.system import GraphedTrainStep, NeRFSystem
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        hp = Namespace(N_samples=64, N_importance=64, use_disp=False, perturb=1.0, noise_std=0.0, chunk=1024 * 32,
                       loss_type="mse", lr=5e-4, weight_decay=0, decay_step=[100], decay_gamma=0.5, white_back=True)
        system = NeRFSystem(hp)
        system.nerf_coarse.load_state_dict(O.make_params(5, 4.0, 0.2))
        system.nerf_fine.load_state_dict(O.make_params(6, 4.0, 0.2))
        for m in system.models:
            m.mlp_dtype = "bf16"
        system = system.to(dev)
        (opt,), _ = system.configure_optimizers()
        sync = GradSync(system.models, force=True)
        batch = {"rays": O.make_rays(1, 256, "blender").to(dev), "rgbs": torch.rand(256, 3, device=dev)}
        for in_graph in (False, True):          # [fwd+bwd graph] -> eager all-reduce -> [optimizer graph]  |  one graph
            stepper = GraphedTrainStep(system, opt, grad_sync=sync, warmup=2, sync_in_graph=in_graph)
            ls = [stepper(batch)["loss"].item() for _ in range(8)]
            assert stepper.graph is not None and (stepper.graph_opt is None) == in_graph
            assert all(torch.isfinite(torch.tensor(ls))) and min(ls[-3:]) < ls[0], (in_graph, ls)
    finally:
        dist.destroy_process_group()


def test_stock_ddp_world1_on_flat_buffer_grads(dev):
    """INTEGRATION.md "DDP works as is" (the reference gets DDP from Lightning, train.py:174-175): stock
    torch.nn.parallel.DistributedDataParallel around NeRFSystem — its autograd hooks, bucket copies and RCCL all-reduce on top
    of parameter gradients that are views of the flat buffer the dW-reduce kernel wrote.  World size 1: the averaged gradients
    must equal the plain ones."""
    import os
    import socket
    import torch.distributed as dist
    from argparse import Namespace
    from nerf_pl_amd.system import NeRFSystem
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    hp = Namespace(N_samples=64, N_importance=64, use_disp=False, perturb=0.0, noise_std=0.0, chunk=1024 * 32,
                   loss_type="mse", lr=5e-4, weight_decay=0, decay_step=[100], decay_gamma=0.5, white_back=True, optimizer="adam",
                   flat_optimizer=False)
    batch = {"rays": O.make_rays(1, 200, "blender").to(dev), "rgbs": torch.rand(200, 3, generator=torch.Generator().manual_seed(0)).to(dev)}

    def build():
        system = NeRFSystem(hp)
        system.nerf_coarse.load_state_dict(O.make_params(5, 4.0, 0.2))
        system.nerf_fine.load_state_dict(O.make_params(6, 4.0, 0.2))
        for m in system.models:
            m.mlp_dtype = "fp32"
        return system.to(dev)
    plain = build()
    plain.loss(plain(batch["rays"]), batch["rgbs"]).backward()
    want = {n: p.grad.clone() for n, p in plain.named_parameters()}
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        system = build()
        ddp = torch.nn.parallel.DistributedDataParallel(system, device_ids=[dev.index or 0])
        for _ in range(2):                                   # twice: the second backward accumulates into / rebuilds bucket views
            system.zero_grad(set_to_none=True)
            system.loss(ddp(batch["rays"]), batch["rgbs"]).backward()
        torch.cuda.synchronize()
        for n, p in system.named_parameters():
            assert p.grad is not None and torch.allclose(p.grad, want[n], rtol=1e-5, atol=1e-9), n
        opt = torch.optim.Adam(system.parameters(), lr=5e-4)
        opt.step()
    finally:
        dist.destroy_process_group()


def test_graphed_step_recaptures_on_lr_change(dev):
    """A scheduler step changes the learning rate (a captured kernel argument): the next call must re-capture."""
    from argparse import Namespace
    from nerf_pl_amd.system import GraphedTrainStep, NeRFSystem
    hp = Namespace(N_samples=64, N_importance=64, use_disp=False, perturb=1.0, noise_std=0.0, chunk=1024 * 32, loss_type="mse",
                   lr=5e-4, weight_decay=0, decay_step=[1], decay_gamma=0.5, white_back=True)
    system = NeRFSystem(hp)
    system.nerf_coarse.load_state_dict(O.make_params(5, 4.0, 0.2))      # seeded, non-dead density (see test_rays.py)
    system.nerf_fine.load_state_dict(O.make_params(6, 4.0, 0.2))
    for m in system.models:
        m.mlp_dtype = "bf16"
    system = system.to(dev)
    (opt,), (sched,) = system.configure_optimizers()
    stepper = GraphedTrainStep(system, opt, warmup=1)
    batch = {"rays": O.make_rays(1, 128, "blender").to(dev), "rgbs": torch.rand(128, 3, device=dev)}
    for _ in range(4):
        stepper(batch)
    g0, w0 = stepper.graph, system.nerf_fine.sigma.weight.detach().clone()
    assert g0 is not None and stepper.captured_lr == 5e-4
    sched.step()                                            # epoch boundary: lr 5e-4 -> 2.5e-4 (MultiStepLR, README.md:192)
    stepper(batch)
    assert stepper.graph is not g0 and stepper.captured_lr == 2.5e-4
    assert not torch.equal(system.nerf_fine.sigma.weight.detach(), w0)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16_f8"])
def test_pack_weights_train_equals_separate_packs(dev, dtype):
    """nerfhip_mlp_pack_weights_train (forward + W^T image in one launch, what a training forward uses) writes byte for
    byte what nerfhip_mlp_pack_weights and nerfhip_mlp_pack_weights_bwd write."""
    from nerf_pl_amd.models.nerf import NeRF
    m = NeRF()
    m.load_state_dict(O.make_params(11, 4.0, 0.2))
    m = m.to(dev)
    fwd = m.packed_weights(dtype).clone()
    bwd = m.packed_weights_bwd(dtype).clone()
    m._packed_cache.pop(dtype), m._packed_cache.pop(("bwd", dtype))        # fresh (uninitialised) buffers
    f2, b2 = m.packed_weights_train(dtype)
    assert torch.equal(fwd, f2) and torch.equal(bwd, b2)


def test_loss_backward_scales_with_upstream_gradient(dev):
    """MSELoss backward honours the gradient flowing into the loss (loss scaling), through its single fused multiply."""
    from nerf_pl_amd import ops
    g = torch.Generator().manual_seed(3)
    c = torch.rand(64, 3, generator=g).to(dev).requires_grad_(True)
    f = torch.rand(64, 3, generator=g).to(dev).requires_grad_(True)
    t = torch.rand(64, 3, generator=g).to(dev)
    loss, _ = ops.mse_psnr(c, f, t)
    (loss * 0.25).backward()
    ref_c = 0.25 * 2 * (c.detach() - t) / t.numel()
    ref_f = 0.25 * 2 * (f.detach() - t) / t.numel()
    assert torch.allclose(c.grad, ref_c, rtol=1e-6, atol=1e-9) and torch.allclose(f.grad, ref_f, rtol=1e-6, atol=1e-9)
    c2 = c.detach().clone().requires_grad_(True)
    loss2, _ = ops.mse_psnr(c2, None, t)                      # coarse-only (N_importance = 0)
    loss2.backward()
    assert torch.allclose(c2.grad, 2 * (c2.detach() - t) / t.numel(), rtol=1e-6, atol=1e-9)


def test_graphed_step_draws_a_fresh_batch_every_replay(dev):
    """GraphedTrainStep(batch_source=RayStore.sample): the batch is drawn INSIDE the captured step (randint + gen_rays + gather),
    and every replay draws a different one (torch's graph-safe Philox state advances per replay)."""
    from argparse import Namespace
    from nerf_pl_amd.rays import RayStore
    from nerf_pl_amd.system import GraphedTrainStep, NeRFSystem
    hp = Namespace(N_samples=16, N_importance=16, use_disp=False, perturb=1.0, noise_std=0.0, chunk=1024 * 32,
                   loss_type="mse", lr=5e-4, weight_decay=0, decay_step=[100], decay_gamma=0.5, white_back=True)
    system = NeRFSystem(hp)
    system.nerf_coarse.load_state_dict(O.make_params(5, 4.0, 0.2))
    system.nerf_fine.load_state_dict(O.make_params(6, 4.0, 0.2))
    for m in system.models:
        m.mlp_dtype = "bf16_f8"
    system = system.to(dev)
    (opt,), _ = system.configure_optimizers()
    g = torch.Generator().manual_seed(0)
    poses = torch.eye(4)[:3].repeat(3, 1, 1)
    poses[:, 2, 3] = torch.tensor([4.0, 4.2, 3.8])
    store = RayStore(poses.to(dev), torch.rand(3, 32, 32, 3, generator=g).to(dev), 32, 32, 40.0, 2.0, 6.0)
    seen = {}

    def source():
        b = store.sample(64)
        seen["rays"] = b["rays"]           # after capture: the graph's own tensor, rewritten by every replay
        return b
    stepper = GraphedTrainStep(system, opt, warmup=2, batch_source=source)
    drawn, losses = [], []
    for _ in range(7):
        losses.append(stepper()["loss"].item())
        drawn.append(seen["rays"].clone())
    assert stepper.graph is not None
    for i in range(3, 7):                  # replays (calls 4..7): each differs from the one before
        assert not torch.equal(drawn[i], drawn[i - 1]), i
    assert all(torch.isfinite(torch.tensor(losses)))
    with pytest.raises(ValueError):
        stepper({"rays": drawn[0], "rgbs": torch.zeros(64, 3, device=dev)})


# ---- round 6: xyz_encoding_final folded out of the saved tensors (csrc/mlp_layout.h kDwJobs, mlp_bwd_fold_kernel) -------------------
def _fold_case(dev, dtype, n=1000):
    from nerf_pl_amd import ops
    g = torch.Generator().manual_seed(11)
    p = O.make_params(5, 3.0, 0.1)
    (m,), _ = build_models([p], dev, dtype)
    rays = torch.cat([torch.rand(n, 3, generator=g) - 0.5, torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1),
                      torch.full((n, 1), 2.0), torch.full((n, 1), 6.0)], 1).to(dev)
    z = (2.0 + 4.0 * torch.rand(n, 1, generator=g)).to(dev)
    acts = ops.alloc_acts(n, dtype, dev)
    raw = ops.mlp_fwd_rays(rays, z, m.packed_weights(dtype), False, dtype, save=acts)
    g_out = torch.randn(n, 4, generator=g).to(dev)
    return ops, m, p, acts, raw, g_out


def _poison_slots(buf, tile_bytes, first_piece, n_pieces, piece_bytes):
    """0xFF (a NaN in bf16, fp32, e4m3 and e5m2 alike) over the `n_pieces` slab / pair slots from `first_piece` of every tile block"""
    v = buf[:buf.numel() // tile_bytes * tile_bytes].view(-1, tile_bytes)
    v[:, first_piece * piece_bytes:(first_piece + n_pieces) * piece_bytes] = 0xFF


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16_f8"])
def test_final_layer_fold_nobody_reads_f_or_its_gradient(dev, dtype):
    """The forward does not save f = xyz_encoding_final(h8) and the chain does not store dL/df (nerf.py:70,116: a linear layer
    without activation): their slots in the saved-activation and dY blocks are never written and never read.  Filled with NaNs —
    before the backward in X, between the chain and the weight-gradient launch in dY — the 24 gradients come out bit-identical."""
    ops, m, _, acts, raw, g_out = _fold_case(dev, dtype)
    pb = m.packed_weights_bwd(dtype)
    ws = {}
    gw, gb, flat = ops.mlp_bwd(g_out, raw, pb, acts, dtype, workspace=ws)
    ref = flat.clone()
    assert torch.isfinite(ref).all() and ref.abs().max().item() > 0
    # slot geometry (csrc/mlp_layout.h): bf16 / fp32 slabs kActFeat = 134 .. 149 of the 158 (+ 9 KiB of gates), kDyFeat = 10 .. 25 of 156;
    # e4m3 / e5m2 pair pieces 67 .. 74 of 79 (+ 9 gate pieces + 1 scale piece), 5 .. 12 of 78 (+ 1 scale piece)
    if dtype == "bf16_f8":
        geo = ((79 + 9 + 1) * 1024, 67, 8, 1024), ((78 + 1) * 1024, 5, 8, 1024)
    else:
        sb = 2048 if dtype == "fp32" else 1024
        geo = (158 * sb + 9 * 1024, 134, 16, sb), (156 * sb, 10, 16, sb)
    _poison_slots(acts, *geo[0])
    ops.mlp_bwd(g_out, raw, pb, acts, dtype, phases=1, workspace=ws)           # chain: writes dY (not dL/df)
    _poison_slots(ws["dys"], *geo[1])
    _, _, flat2 = ops.mlp_bwd(g_out, raw, pb, acts, dtype, phases=6, workspace=ws)   # dW + reduce + fold on the poisoned buffers
    assert torch.equal(flat2, ref)


def test_final_layer_fold_equals_the_direct_products_fp32(dev):
    """dW_dir[:, :256] = sum_p dY_dir f^T, dW_final = sum_p (W_dx^T dY_dir) h8^T, db_final = sum_p W_dx^T dY_dir in float64 from the
    oracle's own activations, against what the dir job's G and mlp_bwd_fold_kernel produce (fp32 path): the same sums,
    re-associated — 1e-4 of each tensor's max |g| (the bound of the other fp32 gradients is 2e-4), measured value printed."""
    ops, m, p, acts, raw, g_out = _fold_case(dev, "fp32", n=2000)
    gw, gb, _ = ops.mlp_bwd(g_out, raw, m.packed_weights_bwd("fp32"), acts, "fp32")
    # float64 autograd through the oracle's MLP on the same points
    g = torch.Generator().manual_seed(11)
    n = 2000
    o = torch.rand(n, 3, generator=g) - 0.5
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    z = 2.0 + 4.0 * torch.rand(n, 1, generator=g)
    x = torch.cat([O.posenc(o + d * z, 10), O.posenc(d, 4)], 1).double()
    p64 = {k: v.double().clone().requires_grad_(True) for k, v in p.items()}
    out = O.mlp_forward(p64, x)
    (out * g_out.cpu().double()).sum().backward()
    names = list(m.state_dict().keys())
    for idx in (8, 9):                                     # xyz_encoding_final, dir_encoding
        for kind, got in (("weight", gw[idx]), ("bias", gb[idx])):
            name = [k for k in names if k.endswith(kind)][idx]
            r = p64[name].grad
            err = (got.cpu().double() - r).abs().max().item()
            print("fold vs float64 autograd: %s max err / max |g| = %.2e" % (name, err / r.abs().max().item()))
            assert err <= 1e-4 * r.abs().max().item() + 1e-9, (name, err, r.abs().max().item())


@pytest.mark.parametrize("B,S", [(32, 64), (8, 192), (64, 32)])
def test_dw_regenerated_encodings_are_the_saved_ones_bf16(dev, B, S):
    """nerfhip_mlp_bwd_multi_rays (bf16): the weight-gradient launch forms embedding_xyz(o + d z) and embedding_dir(d) (nerf.py:21-38,
    rendering.py:186,206-207) — the X operands of xyz_encoding_1, the skip layer and dir_encoding — from the rays and the depths
    instead of reading the six encoding slabs of every saved tile block.  Same arithmetic as the forward, so with those slabs
    NaN-poisoned the 24 gradients are BIT-identical to the ones computed from the saved slabs (one workgroup per job at this size:
    the same split plan, hence the same summation order)."""
    from nerf_pl_amd import ops
    g = torch.Generator().manual_seed(3)
    (m,), _ = build_models([O.make_params(5, 3.0, 0.1)], dev, "bf16")
    rays = O.make_rays(9, B).to(dev)
    z = (2.0 + 4.0 * torch.rand(B, S, generator=g)).sort(-1)[0].to(dev)
    n = B * S
    acts = ops.alloc_acts(n, "bf16", dev)
    raw = ops.mlp_fwd_rays(rays, z, m.packed_weights("bf16"), False, "bf16", save=acts)
    g_out = torch.randn(n, 4, generator=g).to(dev)
    pb = m.packed_weights_bwd("bf16")
    ((_, _, ref),) = ops.mlp_bwd_multi([(g_out, raw, pb, acts)], "bf16")
    ref = ref.clone()
    assert torch.isfinite(ref).all() and ref.abs().max().item() > 0
    _poison_slots(acts, 158 * 1024 + 9 * 1024, 0, 6, 1024)               # kActEncX = 0 (4 slabs), kActEncD = 4 (2 slabs)
    ((_, _, bad),) = ops.mlp_bwd_multi([(g_out, raw, pb, acts)], "bf16")
    assert not torch.isfinite(bad).all()                                  # (the poison is where the launch reads without the rays)
    ((_, _, got),) = ops.mlp_bwd_multi([(g_out, raw, pb, acts, rays, z)], "bf16")
    assert torch.equal(got, ref)
