"""Calibration of the PSNR gate (tests/test_gpu_bf16.py): run-to-run noise of the fp32 path (different jitter seed) vs the
gap of the reduced-precision configurations on the analytic scene; reduced-precision weights are also evaluated through the
fp32 forward to separate training effects from evaluation noise.   python tests/tools/psnr_gate_probe.py [--steps 1000]"""
import argparse
import json
import os
import sys
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import analytic_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--dtypes", default="fp32,fp32b,bf16")
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--decay-at", type=int, default=0, help="multiply the lr by --decay-gamma from this step on (0 = never)")
    ap.add_argument("--decay-gamma", type=float, default=0.1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "psnr_gate_probe.json"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from nerf_pl_amd.inference import batched_inference
    from nerf_pl_amd.models import NeRF
    from nerf_pl_amd.system import NeRFSystem
    S, N, B = 64, 64, 1024
    rays, rgbs = analytic_scene(200000, 1, dev)
    rays_val, rgb_val = analytic_scene(8192, 2, dev)
    torch.manual_seed(0)
    init = [NeRF().state_dict(), NeRF().state_dict()]
    res = {}
    for tag in a.dtypes.split(","):
        dtype = "fp32" if tag.startswith("fp32") else tag
        hp = Namespace(N_samples=S, N_importance=N, use_disp=False, perturb=1.0, noise_std=0.0, chunk=32768, loss_type="mse",
                       lr=a.lr, weight_decay=0, decay_step=[10 ** 9], decay_gamma=0.5, white_back=True)
        system = NeRFSystem(hp)
        system.nerf_coarse.load_state_dict(init[0])
        system.nerf_fine.load_state_dict(init[1])
        for m in system.models:
            m.mlp_dtype = dtype
        system = system.to(dev)
        (opt,), _ = system.configure_optimizers()
        torch.manual_seed(99 if tag == "fp32b" else 1234)
        perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
        curve, curve32 = {}, {}
        for step in range(1, a.steps + 1):
            idx = perm[((step - 1) * B) % (rays.shape[0] - B):][:B]
            if a.decay_at and step == a.decay_at:
                for grp in opt.param_groups:
                    grp["lr"] = a.lr * a.decay_gamma
            out = system.training_step({"rays": rays[idx], "rgbs": rgbs[idx]}, step)
            opt.zero_grad(set_to_none=True)
            out["loss"].backward()
            opt.step()
            if step % a.every == 0:
                with torch.no_grad():
                    img = batched_inference(system.models, system.embeddings, rays_val, S, N, False, 32768, True)["rgb_fine"]
                    curve[step] = round((-10 * torch.log10(torch.mean((img - rgb_val) ** 2))).item(), 3)
                    if dtype != "fp32":
                        for m in system.models:
                            m.mlp_dtype = "fp32"
                        img = batched_inference(system.models, system.embeddings, rays_val, S, N, False, 32768, True)["rgb_fine"]
                        curve32[step] = round((-10 * torch.log10(torch.mean((img - rgb_val) ** 2))).item(), 3)
                        for m in system.models:
                            m.mlp_dtype = dtype
        res[tag] = curve
        if curve32:
            res[tag + "_eval_fp32"] = curve32
        print(tag, curve, flush=True)
        if curve32:
            print(tag + "_eval_fp32", curve32, flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
