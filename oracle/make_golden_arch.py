"""Golden vectors of NON-default NeRF shapes, minted by executing the REAL reference (kwea123/nerf_pl @ /root/reference).

    python oracle/make_golden_arch.py     # writes tests/golden/reference_golden_arch.npz   (build container only)

The reference's NeRF takes any D / W / skips / channel counts (models/nerf.py:42-81) and Embedding any number of bands, log- or
linearly spaced (nerf.py:5-19).  For each configuration of ARCHS: NeRF.forward (full and sigma_only) with the gradients of a
fixed linear functional w.r.t. every parameter and the input, and render_rays (perturb = 0, noise_std = 0: the reference's
randn draws are multiplied by 0) with the gradients of the training loss (losses.py:9-14).  Weights = nerf_oracle.make_params
(seed, arch=...), so the fixture stores seeds, inputs and the reference's outputs only.
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

OUT = os.path.join(ROOT, "tests", "golden", "reference_golden_arch.npz")

from oracle.arch_cases import ARCHS, N_I, N_PTS, N_RAYS, S_C  # noqa: E402


def ref_models(nerf, arch, seed):
    out = []
    for k in range(2):
        p = O.make_params(seed + k, 6.0, 0.3, arch=arch)
        m = nerf.NeRF(D=arch["D"], W=arch["W"], in_channels_xyz=arch["in_xyz"], in_channels_dir=arch["in_dir"],
                      skips=list(arch["skips"]))
        m.load_state_dict(p)
        out.append(m)
    return out


def main():
    nerf, rend = ref_shim.load_reference()
    out = {}
    for tag, (kw, seed) in ARCHS.items():
        arch = O.make_arch(**kw)
        g = torch.Generator().manual_seed(seed)
        models = ref_models(nerf, arch, seed)
        c_in = arch["in_xyz"] + arch["in_dir"]
        # --- NeRF.forward on pre-embedded inputs
        x = (torch.rand(N_PTS, c_in, generator=g) * 2 - 1).requires_grad_(True)
        G = torch.randn(N_PTS, 4, generator=g)
        o = models[0](x)
        (o * G).sum().backward()
        out[tag + "/x"] = x.detach().numpy()
        out[tag + "/G"] = G.numpy()
        out[tag + "/out"] = o.detach().numpy()
        out[tag + "/gx"] = x.grad.numpy().copy()
        for name, p in models[0].named_parameters():
            out[tag + "/g/" + name] = p.grad.numpy().copy()
            p.grad = None
        with torch.no_grad():
            out[tag + "/sigma_only"] = models[0](x[:, :arch["in_xyz"]].detach(), sigma_only=True).numpy()
        # --- render_rays + training loss
        rays = O.make_rays(seed, N_RAYS, "blender")
        target = torch.rand(N_RAYS, 3, generator=g)
        embs = [nerf.Embedding(3, arch["n_freq_xyz"], logscale=arch["logscale"]),
                nerf.Embedding(3, arch["n_freq_dir"], logscale=arch["logscale"])]
        res = rend.render_rays(models, embs, rays, S_C, False, 0, 0, N_I, 1000, True, False)     # chunk 1000: several MLP chunks
        loss = ((res["rgb_coarse"] - target) ** 2).mean() + ((res["rgb_fine"] - target) ** 2).mean()
        loss.backward()
        out[tag + "/rays"] = rays.numpy()
        out[tag + "/target"] = target.numpy()
        out[tag + "/loss"] = np.float32(loss.item())
        for k, v in res.items():
            out[tag + "/render/" + k] = v.detach().numpy()
        for mi, m in enumerate(models):
            for name, p in m.named_parameters():
                out[tag + "/rg%d/" % mi + name] = O.grad_digest(p.grad).numpy()       # [sum, l2, first 8, last 8]
        with torch.no_grad():
            tt = rend.render_rays(models, embs, rays, S_C, False, 0, 0, N_I, 1000, True, True)
        for k, v in tt.items():
            out[tag + "/render_tt/" + k] = v.numpy()
        # the oracle restatement must reproduce the reference on the same inputs (pins the oracle for these shapes)
        params = [O.make_params(seed + k, 6.0, 0.3, arch=arch) for k in range(2)]
        with torch.no_grad():
            assert torch.allclose(O.mlp_forward(params[0], x.detach(), arch=arch), o.detach(), rtol=1e-5, atol=1e-6)
            mine = O.render_rays(params, rays, S_C, False, 0, 0, N_I, True, False, arch=arch)
        for k in res:
            assert torch.allclose(mine[k], res[k].detach(), rtol=1e-5, atol=1e-6), (tag, k)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024), len(out), "arrays")


if __name__ == "__main__":
    main()
