"""PSNR-vs-step of the HIP training path: bf16 MFMA vs exact-fp32 MFMA (and, for the first steps, the CPU
oracle) from identical initial weights and identical ray batches.

There is no dataset in this environment (no Blender/lego files, no network), so the scene is procedural:
a fixed "teacher" radiance field (a NeRF with a sharpened density head) rendered by the fp32 HIP path
provides the ground-truth colour of every training / held-out ray.  The question answered is the one
BASELINE.json asks of the bf16 configuration: does PSNR at equal steps stay within 0.1 dB of fp32?

    python tests/tools/psnr_vs_step.py [--steps 2000] [--oracle-steps 20] [--out gpurun_out/psnr_r01.json]
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import nerf_oracle as O  # noqa: E402  (tool, not product)


def psnr(a, b):
    return (-10 * torch.log10(torch.mean((a - b) ** 2))).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--oracle-steps", type=int, default=20)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--n-train", type=int, default=400000)
    ap.add_argument("--n-importance", type=int, default=128)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "psnr_r01.json"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from nerf_pl_amd.inference import batched_inference
    from nerf_pl_amd.models import Embedding, NeRF
    from nerf_pl_amd.system import NeRFSystem

    S, N, B = 64, a.n_importance, a.rays
    emb = [Embedding(3, 10), Embedding(3, 4)]

    # ---- teacher scene -> ground truth -------------------------------------------------------
    teacher = []
    for seed in (777, 778):
        m = NeRF()
        p = O.make_params(seed, 30.0, 0.0)                   # sharp density head ...
        p["rgb.0.weight"] = p["rgb.0.weight"] * 30.0         # ... saturated colours ...
        for i in range(2, 9):                                # ... and a rougher field: rgb std 0.27, opacity 0.68 +- 0.31
            p[f"xyz_encoding_{i}.0.weight"] = p[f"xyz_encoding_{i}.0.weight"] * 1.5
        m.load_state_dict(p)
        m.mlp_dtype = "fp32"
        teacher.append(m.to(dev))
    rays_train = O.make_rays(11, a.n_train, "blender").to(dev)
    rays_val = O.make_rays(12, 16384, "blender").to(dev)
    with torch.no_grad():
        gt_train = batched_inference(teacher, emb, rays_train, S, N, False, 32768, True)["rgb_fine"]
        gt_val = batched_inference(teacher, emb, rays_val, S, N, False, 32768, True)["rgb_fine"]
    print("teacher rendered: mean rgb %.3f, std %.3f" % (gt_train.mean().item(), gt_train.std().item()), flush=True)

    torch.manual_seed(0)
    init = [NeRF().state_dict(), NeRF().state_dict()]               # default nn.Linear init, coarse then fine
    perm = torch.randperm(a.n_train, generator=torch.Generator().manual_seed(3)).to(dev)
    eval_at = sorted({0, 10, 20, 50, 100, 200, 500, 1000, 1500, 2000, 3000, 5000} & set(range(a.steps + 1)) | {a.steps})
    hp = Namespace(N_samples=S, N_importance=N, use_disp=False, perturb=1.0, noise_std=0.0, chunk=32768, loss_type="mse",
                   lr=5e-4, weight_decay=0, decay_step=[10 ** 9], decay_gamma=0.5, white_back=True)

    def run(dtype):
        system = NeRFSystem(hp)
        system.nerf_coarse.load_state_dict(init[0])
        system.nerf_fine.load_state_dict(init[1])
        for m in system.models:
            m.mlp_dtype = dtype
        system = system.to(dev)
        (opt,), _ = system.configure_optimizers()
        torch.manual_seed(1234)                                        # same perturb/u draws for both dtypes
        curve, losses = {}, []
        t0 = time.perf_counter()
        for step in range(a.steps + 1):
            if step in eval_at:
                with torch.no_grad():
                    pred = batched_inference(system.models, emb, rays_val, S, N, False, 32768, True)["rgb_fine"]
                curve[step] = round(psnr(pred, gt_val), 3)
            if step == a.steps:
                break
            idx = perm[(step * B) % (a.n_train - B):][:B]
            out = system.training_step({"rays": rays_train[idx], "rgbs": gt_train[idx]}, step)
            opt.zero_grad(set_to_none=True)
            out["loss"].backward()
            opt.step()
            if step < a.oracle_steps:
                losses.append(out["loss"].item())
        torch.cuda.synchronize()
        return curve, losses, time.perf_counter() - t0

    res = {"config": {"rays_per_step": B, "N_samples": S, "N_importance": N, "lr": 5e-4, "perturb": 1.0, "noise_std": 0.0,
                      "scene": "procedural teacher NeRF (make_params(777/778), sigma gain 30, rgb gain 30, hidden gain 1.5), %d train rays, 16384 held-out"
                               % a.n_train}}
    for dtype in ("fp32", "bf16"):
        curve, losses, dt = run(dtype)
        res[dtype] = {"psnr_val_at_step": curve, "first_losses": [round(x, 6) for x in losses], "wall_s": round(dt, 1)}
        print(dtype, curve, "wall %.1fs" % dt, flush=True)
    res["delta_bf16_minus_fp32_dB"] = {k: round(res["bf16"]["psnr_val_at_step"][k] - res["fp32"]["psnr_val_at_step"][k], 3)
                                       for k in res["fp32"]["psnr_val_at_step"]}

    # ---- CPU oracle on the first steps, same init / batches (different RNG stream for the jitter) ----
    if a.oracle_steps > 0:
        params = [{k: v.clone().requires_grad_(True) for k, v in sd.items()} for sd in init]
        opt = torch.optim.Adam([v for d in params for v in d.values()], lr=5e-4, eps=1e-8)
        ol = []
        for step in range(a.oracle_steps):
            idx = perm[(step * B) % (a.n_train - B):][:B]
            rays, tgt = rays_train[idx].cpu(), gt_train[idx].cpu()
            rng = O.draw_rng(1000 + step, B, S, N, 1.0)
            out = O.render_rays(params, rays, S, False, 1.0, 0, N, True, False, rng=rng)
            loss = O.mse_loss(out, tgt)
            opt.zero_grad()
            loss.backward()
            opt.step()
            ol.append(round(loss.item(), 6))
        res["oracle_cpu_first_losses"] = ol
        print("oracle", ol, flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["delta_bf16_minus_fp32_dB"]))


if __name__ == "__main__":
    main()
