"""PSNR@step of the HIP training path against the pinned CPU oracle (the restatement of the reference's render_rays +
MSELoss + Adam, oracle/nerf_oracle.py) with REPLAYED randomness: the same default-init weights, the same ray batches and
the same perturb / u draws (tests/helpers.ReplayRNG) go to the oracle (torch-CPU fp32 autograd), HIP fp32, HIP bf16 and
HIP bf16_f8; held-out PSNR is evaluated at the same checkpoints (README.md:75-83 recipe at reduced batch: 256 rays,
64+64 samples, Adam 5e-4, white background, noise_std 0).  Run ON THE GPU BOX (the oracle runs on its host cores):

    python tests/tools/psnr_vs_oracle.py [--steps 250] [--every 25] [--out gpurun_out/psnr_vs_oracle.json]"""
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
from tests.helpers import ReplayRNG, analytic_scene  # noqa: E402


def run_oracle(a, init, draws, ckpts, rays_c, rgbs_c, rays_vc, rgb_vc, S, N, B):
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    params = [{k: v.clone().requires_grad_(True) for k, v in sd.items()} for sd in init]
    opt = torch.optim.Adam([v for d in params for v in d.values()], lr=5e-4, eps=1e-8)
    curve, losses, t0 = {}, [], time.time()
    for step in range(1, a.steps + 1):
        sl = slice((step - 1) * B, step * B)
        out = O.render_rays(params, rays_c[sl], S, False, 1.0, 0.0, N, True, False, rng=draws[step - 1])
        loss = O.mse_loss(out, rgbs_c[sl])
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if step in ckpts:
            with torch.no_grad():
                img = torch.cat([O.render_rays(params, rays_vc[i:i + 1024], S, False, 0, 0, N, True, True)["rgb_fine"]
                                 for i in range(0, rays_vc.shape[0], 1024)])
            curve[step] = O.psnr(img, rgb_vc).item()
    print("oracle: %.0f s on %d threads" % (time.time() - t0, torch.get_num_threads()), {k: round(v, 3) for k, v in curve.items()}, flush=True)
    return {"psnr": curve, "loss": losses}


def oracle_only(a):
    """The oracle's curve on this machine's CPU (no GPU needed): same scene, init, batches and draws as main()."""
    from nerf_pl_amd.models import NeRF
    S, N, B = 64, 64, a.rays
    cpu = torch.device("cpu")
    rays_c, rgbs_c = analytic_scene(a.steps * B, 1, cpu)
    rays_vc, rgb_vc = analytic_scene(4096, 2, cpu)
    torch.manual_seed(0)
    init = [NeRF().state_dict(), NeRF().state_dict()]
    draws = [O.draw_rng(10000 + s, B, S, N, 1.0) for s in range(a.steps)]
    ckpts = [s for s in range(a.every, a.steps + 1, a.every)]
    r = run_oracle(a, init, draws, ckpts, rays_c, rgbs_c, rays_vc, rgb_vc, S, N, B)
    with open(a.oracle_json, "w") as fh:
        json.dump({"key": "steps=%d,every=%d,rays=%d" % (a.steps, a.every, B), "psnr": r["psnr"], "loss": r["loss"],
                   "what": "CPU oracle (oracle/nerf_oracle.py, torch-CPU fp32) trained by tests/tools/psnr_vs_oracle.py --oracle-only"}, fh, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--every", type=int, default=25)
    ap.add_argument("--rays", type=int, default=256)
    ap.add_argument("--dtypes", default="fp32,bf16,bf16_f8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "psnr_vs_oracle.json"))
    ap.add_argument("--oracle-json", default=os.path.join(ROOT, "profiles", "r02_oracle_curve.json"),
                    help="cache of the oracle's curve (it is a pure function of the seeds): computed and written when missing "
                         "(`--oracle-only` does just that, on any CPU), reused otherwise")
    ap.add_argument("--oracle-only", action="store_true")
    a = ap.parse_args()
    if a.oracle_only:
        return oracle_only(a)
    dev = torch.device("cuda:0")
    from nerf_pl_amd.inference import batched_inference
    from nerf_pl_amd.models import NeRF, rendering
    from nerf_pl_amd.system import NeRFSystem
    S, N, B = 64, 64, a.rays
    rays_d, rgbs_d = analytic_scene(a.steps * B, 1, dev)
    rays_v, rgb_v = analytic_scene(4096, 2, dev)
    rays_c, rgbs_c, rays_vc, rgb_vc = rays_d.cpu(), rgbs_d.cpu(), rays_v.cpu(), rgb_v.cpu()
    torch.manual_seed(0)
    init = [NeRF().state_dict(), NeRF().state_dict()]
    draws = [O.draw_rng(10000 + s, B, S, N, 1.0) for s in range(a.steps)]       # per-step perturb / noise / u tensors
    ckpts = [s for s in range(a.every, a.steps + 1, a.every)]
    res = {}

    # ---- the oracle (CPU) -------------------------------------------------------------------------------------------
    key = "steps=%d,every=%d,rays=%d" % (a.steps, a.every, B)
    cached = json.load(open(a.oracle_json)) if os.path.exists(a.oracle_json) else {}
    if cached.get("key") == key:
        res["oracle_cpu_fp32"] = {"psnr": {int(k): v for k, v in cached["psnr"].items()}, "loss": cached["loss"]}
        print("oracle curve from", a.oracle_json, flush=True)
    else:
        res["oracle_cpu_fp32"] = run_oracle(a, init, draws, ckpts, rays_c, rgbs_c, rays_vc, rgb_vc, S, N, B)

    # ---- the HIP path ---------------------------------------------------------------------------------------------
    hp = Namespace(N_samples=S, N_importance=N, use_disp=False, perturb=1.0, noise_std=0.0, chunk=32768, loss_type="mse",
                   lr=5e-4, weight_decay=0, decay_step=[10 ** 9], decay_gamma=0.5, white_back=True)
    order = ["perturb_rand", "noise_coarse", "u", "noise_fine"]
    for dt in a.dtypes.split(","):
        system = NeRFSystem(hp)
        system.nerf_coarse.load_state_dict(init[0])
        system.nerf_fine.load_state_dict(init[1])
        for m in system.models:
            m.mlp_dtype = dt
        system = system.to(dev)
        (opt,), _ = system.configure_optimizers()
        curve, losses = {}, []
        for step in range(1, a.steps + 1):
            sl = slice((step - 1) * B, step * B)
            replay = ReplayRNG(draws[step - 1], order, dev)
            saved, rendering.torch = rendering.torch, replay
            try:
                out = system.training_step({"rays": rays_d[sl], "rgbs": rgbs_d[sl]}, step)
            finally:
                rendering.torch = saved
            assert not replay.q
            opt.zero_grad(set_to_none=True)
            out["loss"].backward()
            opt.step()
            losses.append(out["loss"].item())
            if step in ckpts:
                with torch.no_grad():
                    img = batched_inference(system.models, system.embeddings, rays_v, S, N, False, 32768, True)["rgb_fine"]
                curve[step] = (-10 * torch.log10(torch.mean((img - rgb_v) ** 2))).item()
        res["hip_" + dt] = {"psnr": curve, "loss": losses}
        ref = res["oracle_cpu_fp32"]["psnr"]
        print("hip %s:" % dt, {k: round(v, 3) for k, v in curve.items()},
              " max |dPSNR| vs oracle %.3f dB, first-10-loss max rel diff %.2e"
              % (max(abs(curve[k] - ref[k]) for k in ref),
                 max(abs(x - y) / y for x, y in zip(losses[:10], res["oracle_cpu_fp32"]["loss"][:10]))), flush=True)
    summary = {k: {"max_abs_dpsnr_vs_oracle": round(max(abs(v["psnr"][s] - res["oracle_cpu_fp32"]["psnr"][s]) for s in v["psnr"]), 4),
                   "dpsnr_by_step": {s: round(v["psnr"][s] - res["oracle_cpu_fp32"]["psnr"][s], 4) for s in v["psnr"]}}
               for k, v in res.items() if k != "oracle_cpu_fp32"}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump({"recipe": "analytic scene; %d rays/step x (64+64); Adam 5e-4; perturb=1, noise_std=0, white background; identical "
                             "init, batches and RNG draws for every path; PSNR on 4096 held-out rays" % B,
                   "summary": summary, "curves": res}, fh, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
