"""Mint golden vectors by executing the REAL reference (kwea123/nerf_pl @ /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python oracle/make_golden.py          # writes tests/golden/reference_golden.npz

The reference's `models/nerf.py` and `models/rendering.py` are imported unmodified
(`oracle/ref_shim.py`).  Two pieces of instrumentation wrap them from the outside:
  * the `torchsearchsorted` stand-in records every (cdf, u) -> inds call, which is how the
    bit-exact index vectors are captured at the reference's own call site (rendering.py:42);
  * `torch.rand` / `torch.randn` seen by the reference's rendering module are replaced by a
    replay queue, so that the 3-4 RNG draws per call are known tensors that
    the oracle and the HIP path can be fed verbatim.
Weights come from `nerf_oracle.make_params(seed)` (numpy PCG64) and are loaded into the
reference modules with `load_state_dict`, so fixtures only need to store the seed.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import nerf_oracle as O  # noqa: E402
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "reference_golden.npz")
OUT_GRADS = os.path.join(ROOT, "tests", "golden", "reference_golden_grads.npz")   # gr3: all 48 gradient tensors in full


class _ReplayTorch:
    """Proxy for the `torch` name inside the reference rendering module: rand/randn pop
    pre-drawn tensors (checking the requested shape), everything else is real torch."""

    def __init__(self):
        self.queue = []

    def push(self, kind, t):
        self.queue.append((kind, t))

    def _pop(self, kind, shape):
        assert self.queue, f"reference drew an unexpected {kind}{tuple(shape)}"
        k, t = self.queue.pop(0)
        assert k == kind and tuple(t.shape) == tuple(shape), (k, kind, t.shape, shape)
        return t.clone()

    def rand(self, *shape, **kw):
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        return self._pop("rand", shape)

    def randn(self, *shape, **kw):
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        return self._pop("randn", shape)

    def __getattr__(self, name):
        return getattr(torch, name)


def build_reference():
    nerf, rend = ref_shim.load_reference()
    # record searchsorted calls at the reference's call site
    calls = []
    real = rend.searchsorted

    def recording(a, v, out=None, side="left"):
        r = real(a, v, out=out, side=side)
        calls.append((a.detach().clone(), v.detach().clone(), r.detach().clone(), side))
        return r

    rend.searchsorted = recording
    replay = _ReplayTorch()
    rend.torch = replay
    return nerf, rend, calls, replay


def ref_model(nerf, params):
    m = nerf.NeRF()
    m.load_state_dict(params)
    return m


def main():
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    nerf, rend, calls, replay = build_reference()
    G = {}
    GFULL = {}

    # ---------------------------------------------------------------- 1. Embedding (nerf.py:21-38)
    g = torch.Generator().manual_seed(101)
    x = (torch.rand(61, 3, generator=g) * 12 - 6)
    x = torch.cat([x, torch.tensor([[0.0, -0.0, 1e-30], [6.0, -6.0, 3.14159274], [1e-3, 0.5, -0.25]]),
                   torch.rand(32, 3, generator=g) * 2 - 1], 0).contiguous()
    G["emb_x"] = x
    G["emb_out63"] = nerf.Embedding(3, 10)(x)
    G["emb_out27"] = nerf.Embedding(3, 4)(x)
    assert torch.equal(O.posenc(x, 10), G["emb_out63"]) and torch.equal(O.posenc(x, 4), G["emb_out27"])

    # ---------------------------------------------------------------- 2. NeRF.forward (nerf.py:83-124)
    MLP_SEED = 11
    p = O.make_params(MLP_SEED)
    pts = torch.rand(96, 3, generator=g) * 6 - 3
    dirs = torch.nn.functional.normalize(torch.randn(96, 3, generator=g), dim=-1)
    xin = torch.cat([nerf.Embedding(3, 10)(pts), nerf.Embedding(3, 4)(dirs)], 1).contiguous()
    m = ref_model(nerf, p)
    with torch.no_grad():
        G["mlp_seed"] = torch.tensor(MLP_SEED)
        G["mlp_x"] = xin
        G["mlp_out"] = m(xin)
        G["mlp_sigma"] = m(xin[:, :63], sigma_only=True)

    # ---------------------------------------------------------------- 3/4. sample_pdf + searchsorted
    B, M = 40, 62
    bins = torch.sort(torch.rand(B, M + 1, generator=g) * 4 + 2, -1)[0]
    w = torch.rand(B, M, generator=g)
    w[0] = 0.0                                   # all-zero weights
    w[1] = 0.0; w[1, 17] = 1.0                   # single spike
    w[2] = 0.25                                  # ties: uniform pdf
    w[3] = 0.0; w[3, 0] = 1.0                    # mass in first bin
    w[4] = 0.0; w[4, M - 1] = 1.0                # mass in last bin
    w[5] = 0.0; w[5, 10] = 0.5; w[5, 50] = 0.5   # two spikes, zero-weight bins between
    w[6] = w[6] * 1e-7                           # everything below eps
    w[7, ::2] = 0.0                              # alternating zeros
    w[8:16] = w[8:16] ** 8                       # peaky
    G["sp_bins"], G["sp_w"] = bins, w
    for N_i in (64, 128):
        calls.clear()
        G[f"sp_det{N_i}"] = rend.sample_pdf(bins, w, N_i, det=True)
        a, v, r, side = calls[-1]
        assert side == "right"
        G[f"ss_det{N_i}_cdf"], G[f"ss_det{N_i}_u"], G[f"ss_det{N_i}_inds"] = a, v, r
    u = torch.rand(B, 128, generator=g)
    u[:, 0] = 0.0
    u[:, 1] = 1.0 - 2 ** -24                      # largest fp32 below 1 (torch.rand's max)
    u[:, 2] = 0.5
    # queries landing exactly ON cdf knots (tie handling of side='right')
    cdf_ref = O.pdf_to_cdf(w)
    u[:, 3] = cdf_ref[:, 5]
    u[:, 4] = cdf_ref[:, 31]
    u[:, 5] = cdf_ref[:, 62].clamp(max=1 - 2 ** -24)
    calls.clear()
    replay.push("rand", u)
    G["sp_rand_u"] = u
    G["sp_rand128"] = rend.sample_pdf(bins, w, 128, det=False)
    a, v, r, side = calls[-1]
    G["ss_rand_cdf"], G["ss_rand_inds"] = a, r
    assert torch.equal(v, u)

    # ---------------------------------------------------------------- 5. render_rays configurations
    #   name: (ray kind, B, S_c, N_i, use_disp, perturb, noise_std, white_back, test_time, sigma_gain, sigma_bias)
    cases = {
        "c1_coarse_only": ("blender", 48, 64, 0, False, 0, 0, True, False, 1.0, 0.0),
        "c2_64_64_train": ("blender", 48, 64, 64, False, 1.0, 0.0, True, False, 1.0, 0.0),
        "c3_64_128_train": ("blender", 40, 64, 128, False, 1.0, 1.0, True, False, 1.0, 0.0),
        "c3_64_128_test": ("blender", 40, 64, 128, False, 0, 0, True, True, 1.0, 0.0),
        "c4_ndc_noise": ("ndc", 48, 64, 64, False, 1.0, 1.0, False, False, 1.0, 0.0),
        "disp_32_32": ("blender", 33, 32, 32, True, 0.5, 0.0, False, False, 1.0, 0.0),
        "peaky_64_128": ("blender", 40, 64, 128, False, 0, 0, True, False, 40.0, 1.0),
        "peaky_test": ("blender", 40, 64, 128, False, 0, 0, False, True, 40.0, 1.0),
        "ragged_7rays": ("blender", 7, 24, 40, False, 1.0, 1.0, True, False, 8.0, 0.5),
    }
    emb = [nerf.Embedding(3, 10), nerf.Embedding(3, 4)]
    names = []
    for ci, (name, cfg) in enumerate(cases.items()):
        kind, B, S_c, N_i, disp, pert, nstd, wb, tt, sg, sb = cfg
        seed = 1000 + ci
        pc, pf = O.make_params(seed, sg, sb), O.make_params(seed + 500, sg, sb)
        rays = O.make_rays(seed, B, kind)
        rng = O.draw_rng(seed, B, S_c, N_i, pert)
        for key in ("perturb_rand", "noise_coarse", "u", "noise_fine"):
            if key in rng:
                replay.push("rand" if key in ("perturb_rand", "u") else "randn", rng[key])
        with torch.no_grad():
            res = rend.render_rays([ref_model(nerf, pc), ref_model(nerf, pf)], emb, rays, S_c, disp,
                                   pert, nstd, N_i, 1024 * 32, wb, test_time=tt)
        assert not replay.queue, (name, len(replay.queue))
        G[f"rr_{name}_cfg"] = torch.tensor([{"blender": 0, "ndc": 1}[kind], B, S_c, N_i, int(disp), pert, nstd,
                                            int(wb), int(tt), sg, sb, seed], dtype=torch.float64)
        for k, v in res.items():
            G[f"rr_{name}_{k}"] = v
        names.append(name)
        # sanity: the oracle restatement agrees with the reference on this case
        ores = O.render_rays([pc, pf], rays, S_c, disp, pert, nstd, N_i, wb, tt, rng=rng)
        for k in res:
            err = (ores[k] - res[k]).abs().max().item()
            assert err < 2e-5, (name, k, err)
    G["rr_names"] = np.array(names)

    # ---------------------------------------------------------------- 6. training-loss gradients
    B, S_c, N_i, pert, nstd, wb = 32, 64, 64, 1.0, 1.0, True
    seed = 2000
    pc, pf = O.make_params(seed, 6.0, 0.3), O.make_params(seed + 500, 6.0, 0.3)
    rays = O.make_rays(seed, B, "blender")
    rng = O.draw_rng(seed, B, S_c, N_i, pert)
    tgt = torch.rand(B, 3, generator=g)
    for key in ("perturb_rand", "noise_coarse", "u", "noise_fine"):
        replay.push("rand" if key in ("perturb_rand", "u") else "randn", rng[key])
    mc, mf = ref_model(nerf, pc), ref_model(nerf, pf)
    res = rend.render_rays([mc, mf], emb, rays, S_c, False, pert, nstd, N_i, 1024 * 32, wb)
    loss = torch.nn.functional.mse_loss(res["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(res["rgb_fine"], tgt)
    loss.backward()
    G["gr_cfg"] = torch.tensor([0, B, S_c, N_i, 0, pert, nstd, int(wb), 0, 6.0, 0.3, seed], dtype=torch.float64)
    G["gr_target"] = tgt
    G["gr_loss"] = loss.detach()
    G["gr_rgb_fine"] = res["rgb_fine"].detach()
    for tag, mod in (("c", mc), ("f", mf)):
        for n, prm in mod.named_parameters():
            G[f"gr_{tag}_{n}"] = O.grad_digest(prm.grad)
    # one full gradient tensor per model, for element-wise checks (small ones)
    G["gr_full_c_sigma.weight"] = mc.sigma.weight.grad.clone()
    G["gr_full_f_rgb.0.weight"] = mf.rgb[0].weight.grad.clone()
    G["gr_full_f_xyz_encoding_1.0.bias"] = getattr(mf, "xyz_encoding_1")[0].bias.grad.clone()

    # ---------------------------------------------------------------- 6b. gradients at the other train configurations
    # gr3: BASELINE configs[2] shape (64 + 128 samples, perturb=1, noise_std=0, white background — the README lego recipe);
    # gr4: configs[3] shape (LLFF/NDC rays: near 0, far 1, non-unit directions, noise_std=1, black background, 64 + 64).
    # (fresh generators: nothing above this point may change when cases are added here)
    for prefix, (kind, B, S_c, N_i, pert, nstd, wb, sg, sb, seed) in {
            "gr3": ("blender", 32, 64, 128, 1.0, 0.0, True, 6.0, 0.3, 2100),
            "gr4": ("ndc", 32, 64, 64, 1.0, 1.0, False, 6.0, 0.3, 2200)}.items():
        pc, pf = O.make_params(seed, sg, sb), O.make_params(seed + 500, sg, sb)
        rays = O.make_rays(seed, B, kind)
        rng = O.draw_rng(seed, B, S_c, N_i, pert)
        tgt = torch.rand(B, 3, generator=torch.Generator().manual_seed(seed))
        for key in ("perturb_rand", "noise_coarse", "u", "noise_fine"):
            replay.push("rand" if key in ("perturb_rand", "u") else "randn", rng[key])
        mc, mf = ref_model(nerf, pc), ref_model(nerf, pf)
        res = rend.render_rays([mc, mf], emb, rays, S_c, False, pert, nstd, N_i, 1024 * 32, wb)
        assert not replay.queue
        loss = torch.nn.functional.mse_loss(res["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(res["rgb_fine"], tgt)
        loss.backward()
        G[f"{prefix}_cfg"] = torch.tensor([{"blender": 0, "ndc": 1}[kind], B, S_c, N_i, 0, pert, nstd, int(wb), 0, sg, sb, seed],
                                          dtype=torch.float64)
        G[f"{prefix}_target"] = tgt
        G[f"{prefix}_loss"] = loss.detach()
        G[f"{prefix}_rgb_fine"] = res["rgb_fine"].detach()
        for tag, mod in (("c", mc), ("f", mf)):
            for n, prm in mod.named_parameters():
                G[f"{prefix}_{tag}_{n}"] = O.grad_digest(prm.grad)
        G[f"{prefix}_full_c_sigma.weight"] = mc.sigma.weight.grad.clone()
        G[f"{prefix}_full_f_rgb.0.weight"] = mf.rgb[0].weight.grad.clone()
        G[f"{prefix}_full_f_xyz_encoding_1.0.bias"] = getattr(mf, "xyz_encoding_1")[0].bias.grad.clone()
        G[f"{prefix}_full_f_dir_encoding.0.weight"] = mf.dir_encoding[0].weight.grad.clone()
        if prefix == "gr3":         # configs[2]: EVERY gradient tensor of both models in full (a file of its own: 4.8 MB of fp32)
            for tag, mod in (("c", mc), ("f", mf)):
                for n, prm in mod.named_parameters():
                    GFULL[f"gr3_grad_{tag}_{n}"] = prm.grad.clone()
        # the oracle restatement reproduces these gradients (autograd through nerf_oracle.render_rays)
        opc = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
        opf = {k: v.clone().requires_grad_(True) for k, v in pf.items()}
        ores = O.render_rays([opc, opf], rays, S_c, False, pert, nstd, N_i, wb, False, rng=rng)
        O.mse_loss(ores, tgt).backward()
        for n, prm in mf.named_parameters():
            scale = prm.grad.abs().max().item() + 1e-12
            assert (opf[n].grad - prm.grad).abs().max().item() <= 1e-4 * scale, (prefix, n)

    # ---------------------------------------------------------------- 7. ray geometry (datasets/ray_utils.py), N1
    ru = ref_shim.load_reference_ray_utils()
    for tag, (H, W, focal, pose_seed) in {"blender": (20, 24, 27.7777, 31), "llff": (19, 25, 21.5, 32)}.items():
        c2w = O.make_pose(pose_seed)
        dirs = ru.get_ray_directions(H, W, focal)
        ro, rd = ru.get_rays(dirs, c2w)
        G[f"rg_{tag}_cfg"] = torch.tensor([H, W, focal, pose_seed], dtype=torch.float64)
        G[f"rg_{tag}_dirs"] = dirs
        G[f"rg_{tag}_o"] = ro.contiguous()
        G[f"rg_{tag}_d"] = rd.contiguous()
        assert torch.equal(O.get_ray_directions(H, W, focal), dirs)
        oo, od = O.get_rays(dirs, c2w)
        assert torch.allclose(od, rd, rtol=0, atol=1e-7) and torch.equal(oo, ro)
        if tag == "llff":                                  # forward-facing scenes: NDC with the near plane at 1.0 (llff.py:236-239)
            no, nd = ru.get_ndc_rays(H, W, focal, 1.0, ro, rd)
            G["rg_llff_ndc_o"], G["rg_llff_ndc_d"] = no.contiguous(), nd.contiguous()
            po, pd = O.get_ndc_rays(H, W, focal, 1.0, ro, rd)
            assert torch.equal(po, no) and torch.equal(pd, nd)

    out = {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in G.items()}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024), len(out), "arrays")
    full = {k: v.detach().numpy() for k, v in GFULL.items()}
    np.savez_compressed(OUT_GRADS, **full)
    print("wrote", OUT_GRADS, "%.1f KB" % (os.path.getsize(OUT_GRADS) / 1024), len(full), "arrays")


if __name__ == "__main__":
    main()
