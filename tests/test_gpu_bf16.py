"""GPU: quick guards for the bf16 (and fp8-storage) MFMA configurations, which cannot meet the 1e-4 fp32 parity bound: a
60-second PSNR@step check against GROSS degradation (1.0 dB on the easy scene below — NOT the 0.1 dB criterion of BASELINE.json;
that one, with its 16-seed paired statistics at a 31 dB plateau, is tests/test_gpu_psnr_gate.py) and per-tensor gradient direction
at n >= 1000 points.

The scene is procedural (tests/helpers.analytic_scene: closed-form colours of a soft ball, independent of both the HIP
path and the oracle) and easy enough to pass 25 dB within a few hundred steps, where PSNR is still sensitive.
Every configuration starts from the same default-init weights and consumes the same ray batches and RNG draws."""
from argparse import Namespace

import pytest
import torch

from oracle import nerf_oracle as O
from tests.helpers import analytic_scene, build_models

pytestmark = pytest.mark.gpu

STEPS, DECAY_AT, B, S, N = 900, 600, 1024, 64, 64
EVAL_AT = (700, 800, 900)          # after the learning-rate decay, where PSNR@step is a smooth function of the step


def _train(dtype, dev, rays, rgbs, rays_val, rgb_val, init, jitter_seed=1234):
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
    torch.manual_seed(jitter_seed)                           # the perturb / u draws
    perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
    curve = {}
    for step in range(1, STEPS + 1):
        if step == DECAY_AT:                                 # the recipe's step decay (README.md:75-83), compressed
            for grp in opt.param_groups:
                grp["lr"] = 5e-5
        idx = perm[((step - 1) * B) % (rays.shape[0] - B):][:B]
        out = system.training_step({"rays": rays[idx], "rgbs": rgbs[idx]}, step)
        opt.zero_grad(set_to_none=True)
        out["loss"].backward()
        opt.step()
        if step in EVAL_AT:
            with torch.no_grad():
                img = batched_inference(system.models, system.embeddings, rays_val, S, N, False, 32768, True)["rgb_fine"]
            curve[step] = (-10 * torch.log10(torch.mean((img - rgb_val) ** 2))).item()
    return curve, system


def _dtypes():
    from nerf_pl_amd.ops import _DTYPES
    return ["fp32", "bf16"] + (["bf16_f8"] if "bf16_f8" in _DTYPES else [])


def test_psnr_at_equal_steps_vs_fp32(dev):
    """PSNR@step of the reduced-precision configurations against the fp32 path: same init, same batches, same RNG draws.

    Training is chaotic: a single trajectory pair says little — the CPU oracle run on two hosts (same code, seeds and
    batches) is 0.9 dB apart by step 200 (profiles/archive/r02_oracle_curve.json), two fp32 runs here that differ in the jitter seed
    0.1-0.4 dB, and changing nothing but the split-K partition of the fp32 dW reduction (a different fp32 summation order)
    moved the fp32 curves themselves by 0.03-0.27 dB and the per-seed bf16 - fp32 difference between -1.46 and +0.40 dB.
    A bare `|delta| <= 0.1 dB` on one pair would therefore test the seed, not the arithmetic.  This test is the 60-second
    guard against GROSS degradation (an e4m3 dY cost 1.05 dB over three seeds): over four init/jitter seeds and the three
    post-decay checkpoints, the mean PSNR of every reduced-precision configuration is within 1.0 dB of fp32's (standard error
    of that mean ~0.3 dB); the statistics proper are in profiles/archive/r02_psnr_seeds_final.json (tests/tools/psnr_seeds.py: bf16
    -0.27 +- 0.16 dB, bf16_f8 -0.08 +- 0.07 dB vs fp32 at 42.7 dB, four live seeds).  The noise-free half: the SAME weights
    rendered through the bf16 forward and through the fp32 forward agree to 0.1 dB."""
    from nerf_pl_amd.inference import batched_inference
    rays, rgbs = analytic_scene(200000, 1, dev)
    rays_val, rgb_val = analytic_scene(8192, 2, dev)
    from nerf_pl_amd.models import NeRF
    means, curves, systems = {}, {}, {}
    for seed in (0, 1, 3, 5):          # (2 and 4 are dead-ReLU inits: 7.3 dB in every precision)
        torch.manual_seed(seed)
        init = [NeRF().state_dict(), NeRF().state_dict()]   # default nn.Linear init, coarse then fine (train.py:38-42)
        for dt in _dtypes():
            c, sysm = _train(dt, dev, rays, rgbs, rays_val, rgb_val, init, jitter_seed=1000 + seed)
            curves[(dt, seed)] = c
            means.setdefault(dt, []).append(sum(c[s] for s in EVAL_AT) / len(EVAL_AT))
            if dt == "fp32" and seed == 0:
                systems["fp32"] = sysm
    print("PSNR@step on the analytic scene:", {str(k): {s: round(v, 3) for s, v in c.items()} for k, c in curves.items()})
    mean = {dt: sum(v) / len(v) for dt, v in means.items()}
    print("mean PSNR over seeds and checkpoints:", {k: round(v, 3) for k, v in mean.items()})
    assert min(means["fp32"]) >= 25.0, means["fp32"]       # the scene is in the PSNR-sensitive regime
    for dt in _dtypes()[1:]:
        assert mean[dt] >= mean["fp32"] - 1.0, (dt, mean[dt], mean["fp32"])
    # same weights, two forwards: the fp32-trained model rendered by the bf16 MFMA path
    ms = systems["fp32"].models
    with torch.no_grad():
        ref = batched_inference(ms, systems["fp32"].embeddings, rays_val, S, N, False, 32768, True)["rgb_fine"]
        for m in ms:
            m.mlp_dtype = "bf16"
        low = batched_inference(ms, systems["fp32"].embeddings, rays_val, S, N, False, 32768, True)["rgb_fine"]
    p_ref = (-10 * torch.log10(torch.mean((ref - rgb_val) ** 2))).item()
    p_low = (-10 * torch.log10(torch.mean((low - rgb_val) ** 2))).item()
    print("same fp32-trained weights: fp32 render %.3f dB, bf16 render %.3f dB" % (p_ref, p_low))
    assert abs(p_ref - p_low) <= 0.1, (p_ref, p_low)


# Magnitude gates of the test below (round 4; until then only the direction was asserted).  Measured maxima over the 24 tensors at
# n = 1000 / 4096 (profiles/r04_grad_magnitude.txt): relative L2 error bf16 0.111 / 0.117, bf16_f8 0.142 / 0.176; norm ratio bf16
# 0.982 / 0.987, bf16_f8 0.858 (sigma.bias — ONE element, a sum of 1000 random-sign terms that cancel to 3 % of their size) / 1.061.
REL_L2_MAX = {"bf16": 0.15, "bf16_f8": 0.22}
NORM_RATIO_MAX = {"bf16": 0.04, "bf16_f8": 0.20}


@pytest.mark.parametrize("n", [1000, 4096])
def test_reduced_precision_gradient_direction(dev, n):
    """Per-tensor cosine between the reduced-precision gradients and autograd through the fp32 CPU oracle: >= 0.99 for bf16;
    >= 0.98 for the fp8-storage mode, whose dY operand is e5m2 (2 mantissa bits: zero-mean rounding noise of ~7 % per
    product that averages over the points — this test's random-sign g_out makes the sums cancel heavily, the worst case for
    it: measured minimum 0.9846, a bias gradient; what matters for training is the exponent range — an e4m3 dY, 0.7 % closer
    in cosine here, lost 0.7 dB of PSNR by flushing the small per-point gradients of off-surface samples, profiles/README.md)."""
    g = torch.Generator().manual_seed(n)
    p = O.make_params(21, 3.0, 0.1)
    pts = torch.rand(n, 3, generator=g) * 4 - 2
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    x = torch.cat([O.posenc(pts, 10), O.posenc(dirs, 4)], 1)
    g_out = torch.randn(n, 4, generator=g)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    (O.mlp_forward(pr, x) * g_out).sum().backward()
    worst, worst_rel, worst_ratio = {}, {}, {}
    for dt in _dtypes()[1:]:
        (m,), _ = build_models([p], dev, dt)
        (m(x.to(dev)) * g_out.to(dev)).sum().backward()
        for name, prm in m.named_parameters():
            ref = pr[name].grad
            cos = torch.nn.functional.cosine_similarity(prm.grad.cpu().flatten(), ref.flatten(), dim=0).item()
            rel = (prm.grad.cpu() - ref).norm().item() / (ref.norm().item() + 1e-12)
            ratio = prm.grad.cpu().norm().item() / (ref.norm().item() + 1e-12)
            if cos < worst.get(dt, (1.0, ""))[0]:
                worst[dt] = (cos, name)
            if rel > worst_rel.get(dt, (0.0, ""))[0]:
                worst_rel[dt] = (rel, name)
            if abs(ratio - 1) > abs(worst_ratio.get(dt, (1.0, ""))[0] - 1):
                worst_ratio[dt] = (ratio, name)
            assert cos >= (0.99 if dt == "bf16" else 0.98), (dt, n, name, cos, rel)
    print("worst per-tensor gradient cosine at n=%d:" % n, worst)
    print("worst per-tensor relative L2 error:", worst_rel, " worst norm ratio:", worst_ratio)
    # ... and magnitude, not only direction: relative L2 error and norm ratio of every tensor
    for dt in worst_rel:
        assert worst_rel[dt][0] <= REL_L2_MAX[dt], (dt, n, worst_rel[dt])
        assert abs(worst_ratio[dt][0] - 1) <= NORM_RATIO_MAX[dt], (dt, n, worst_ratio[dt])


# Gates of the test below, per gradient tensor of both models (48).  Measured (r05 call 3, one box): fp32 — cosine 1.00000, relative
# L2 error <= 1e-4 (coarse) / 5e-3 (fine trunk, with the "exact" row total; ATen order: see the test) ; bf16 — cosine >= 0.9926,
# relative L2 <= 0.114, bf16_f8 — cosine >= 0.9922, relative L2 <= 0.125, BOTH except the density head (sigma.weight, sigma.bias):
# its gradient is sum_p g_sigma[p] * h8[p] with h8 >= 0 (post-ReLU) and g_sigma of both signs — samples in front of a surface
# push the density up, samples behind it down — a sum that cancels to ~1 % of its terms, so the 0.4 % rounding of a bf16 forward
# moves it by tens of % (norm ratio 1.46 coarse / 0.82 fine) while its direction holds (cosine 0.9926 / 0.9996).  The head is
# therefore gated on its direction and on its error RELATIVE TO THE TERMS OF THE SUM (sum_p |g_sigma[p]|: `sigma_head_abs`).
# Measured density-head error / terms: fp32 6.1e-6 (coarse) 2.3e-7 (fine); bf16 4.3e-3 / 5.1e-3; bf16_f8 4.1e-3 / 5.3e-3.
# fp32 (the parity arithmetic) holds 1e-3 relative L2 on every tensor but the fine model's trunk (layers 1-8), whose question is
# ~1000x worse conditioned (ulp-level differences of the fine depths: tests/test_oracle_golden.py::test_fine_pass_conditioning;
# measured 1e-3..4.8e-3 there, <= 1e-4 elsewhere).
TIMED_NODE_BOUNDS = {"fp32": dict(cos=0.9999, rel_l2=1e-3, rel_l2_fine_trunk=1.5e-2, loss_rel=1e-5, rgb_abs=1e-4, sigma_head_abs=1e-4),
                     "bf16": dict(cos=0.99, rel_l2=0.15, rel_l2_fine_trunk=0.15, loss_rel=1e-3, rgb_abs=2e-3, sigma_head_abs=1.5e-2),
                     "bf16_f8": dict(cos=0.99, rel_l2=0.16, rel_l2_fine_trunk=0.16, loss_rel=1e-3, rgb_abs=2e-3, sigma_head_abs=1.5e-2)}


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16_f8"])
def test_timed_node_at_benchmark_size_vs_oracle_gradients(dev, dtype):
    """The node bench.py times (models/train_step.render_rays_train: fused launches, one autograd node) in the HEADLINE arithmetic
    (bf16 MFMA, fp32 accumulate) at the HEADLINE size — 1024 rays x (64 + 128) samples, perturb = 1, noise_std = 0, white
    background (BASELINE configs[2]) — against autograd through the fp32 CPU oracle on the same rays, weights and draws: loss,
    rendered colours and every one of the 48 gradient tensors (direction AND magnitude: cosine, relative L2 error).
    The oracle runs the 262,144 points in four ray chunks (the loss is a mean over rays: chunk losses add)."""
    from helpers import fused_draws
    from nerf_pl_amd.models.train_step import render_rays_train
    Bn, Sc, Ni, seed = 1024, 64, 128, 3100
    kw = dict(N_samples=Sc, use_disp=False, perturb=1.0, noise_std=0.0, N_importance=Ni, white_back=True, test_time=False)
    params = [O.make_params(seed, 6.0, 0.3), O.make_params(seed + 500, 6.0, 0.3)]
    rays = O.make_rays(seed, Bn, "blender")
    rng = O.draw_rng(seed, Bn, Sc, Ni, 1.0)
    tgt = torch.rand(Bn, 3, generator=torch.Generator().manual_seed(seed))
    # ---- oracle: fp32 autograd on the CPU, 256 rays at a time
    op = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in params]
    ref_loss, ref_rgb = 0.0, []
    for lo in range(0, Bn, 256):
        sl = slice(lo, lo + 256)
        res = O.render_rays(op, rays[sl], Sc, False, 1.0, 0.0, Ni, True, False, rng={k: v[sl] for k, v in rng.items()})
        part = O.mse_loss(res, tgt[sl]) * (256.0 / Bn)
        part.backward()
        ref_loss += part.item()
        ref_rgb.append(res["rgb_fine"].detach())
    ref_rgb = torch.cat(ref_rgb)
    # ---- the timed node
    ms, emb = build_models(params, dev, dtype)
    res, loss, out3 = render_rays_train(ms, emb, rays.to(dev), tgt.to(dev), Sc, False, 1.0, 0.0, Ni, True, draws=fused_draws(rng, kw, dev))
    loss.backward()
    bd = TIMED_NODE_BOUNDS[dtype]
    loss_rel = abs(loss.item() - ref_loss) / abs(ref_loss)
    rgb_abs = (res["rgb_fine"].cpu() - ref_rgb).abs().max().item()
    rows = []
    for tag, m, o in (("c", ms[0], op[0]), ("f", ms[1], op[1])):
        for n, prm in m.named_parameters():
            g, r = prm.grad.cpu().flatten(), o[n].grad.flatten()
            cos = torch.nn.functional.cosine_similarity(g, r, dim=0).item()
            rel = (g - r).norm().item() / (r.norm().item() + 1e-20)
            rows.append((tag + "." + n, cos, rel, g.norm().item() / (r.norm().item() + 1e-20), r.norm().item()))
    worst_cos, worst_rel = min(rows, key=lambda t: t[1]), max(rows, key=lambda t: t[2])
    print("timed node %s @ 1024 x (64+128) vs fp32 oracle: loss rel err %.2e, rgb_fine max abs err %.2e, worst gradient cosine %.4f (%s), "
          "worst relative L2 error %.4f (%s)" % (dtype, loss_rel, rgb_abs, worst_cos[1], worst_cos[0], worst_rel[2], worst_rel[0]))
    print("  per tensor (name, cosine, relative L2 error, norm ratio, |ref|):")
    for row in rows:
        print("   %-28s %.5f %.4f %.4f %.3e" % row)
    # the non-cancelling scale of the density head's gradient: sum_p |d loss / d sigma_p| of each pass (fp32 launches of the
    # same pieces the node runs)
    from nerf_pl_amd import ops
    m32, _ = build_models(params, dev, "fp32")
    gs = 2.0 / (3 * Bn)
    rd, td = rays.to(dev), tgt.to(dev)
    zc, raw_c = ops.mlp_fwd_rays_coarse(rd, Sc, m32[0].packed_weights("fp32"), False, "fp32", False, 1.0, rng["perturb_rand"].to(dev))
    _, _, _, _, g_raw_c, zf = ops.composite_train_fine_z(raw_c, zc, rd, None, 0.0, True, td, gs, Ni, u=rng["u"].to(dev))
    raw_f = ops.mlp_fwd_rays(rd, zf, m32[1].packed_weights("fp32"), False, "fp32")
    g_raw_f = ops.composite_train(raw_f, zf, rd, None, 0.0, True, td, gs, want_weights=False)[4]
    terms = {"c": g_raw_c[..., 3].abs().sum().item(), "f": g_raw_f[..., 3].abs().sum().item()}
    for tag, m, o in (("c", ms[0], op[0]), ("f", ms[1], op[1])):
        gb, rb = m.sigma.bias.grad.cpu().item(), o["sigma.bias"].grad.item()
        print("  %s density head: sum_p g_sigma = %.3e (oracle %.3e), sum_p |g_sigma| = %.3e -> cancellation x%.0f, error / terms %.2e"
              % (tag, gb, rb, terms[tag], terms[tag] / abs(rb), abs(gb - rb) / terms[tag]))
        assert abs(gb - rb) <= bd["sigma_head_abs"] * terms[tag], (dtype, tag, gb, rb, terms[tag])
    for name, cos, rel, _, _ in rows:
        head = name.endswith(("sigma.weight", "sigma.bias"))
        trunk = name.startswith("f.xyz_encoding_") and not name.startswith("f.xyz_encoding_final")
        assert cos >= bd["cos"] and (head or rel <= (bd["rel_l2_fine_trunk"] if trunk else bd["rel_l2"])), (dtype, name, cos, rel)
    assert loss_rel <= bd["loss_rel"], (loss.item(), ref_loss)
    assert rgb_abs <= bd["rgb_abs"], rgb_abs
