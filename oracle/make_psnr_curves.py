"""Mint PSNR@step training curves by TRAINING THE REAL REFERENCE (kwea123/nerf_pl @ /root/reference).  TEST INFRASTRUCTURE ONLY.

Run in the build container only (the GPU box has no /root/reference):

    python oracle/make_psnr_curves.py [--seeds 12] [--steps 300]     # writes tests/golden/reference_psnr_curves.json

What runs is the reference's own training step (train.py:103-117) on its own arithmetic: the UNMODIFIED `models/nerf.py`
(`NeRF`, `Embedding`), `models/rendering.py` (`render_rays`) and `losses.py` (`MSELoss`) loaded through `oracle/ref_shim.py` (`metrics.py` imports kornia, absent here: the
held-out PSNR is `nerf_oracle.psnr`, the restatement of metrics.py:4-13, -10 log10 of the mean squared error), `torch.optim.Adam(lr=5e-4, eps=1e-8)` as `utils/__init__.py:18-20` builds it, on torch-CPU fp32 — the
README recipe (README.md:75-83: 64 + 64 samples, perturb 1.0, noise_std 0 for Blender scenes, white background) at a reduced
batch of 256 rays of `oracle.scenes.brick_scene` (no dataset offline; closed-form ground truth).  The reference's in-call RNG
draws (rendering.py:203, :152, :39, :152) are replaced from outside by the replay queue of `oracle/make_golden.py`, fed from
`nerf_oracle.draw_rng(7000 * seed + step, ...)`: the HIP path is then trained on the SAME default inits (`torch.manual_seed(seed)`
+ two `NeRF()`), the SAME batches and the SAME draws by `tests/test_gpu_psnr_gate.py`, and the paired PSNR difference on
4,096 held-out rays is the north-star statistic ("PSNR within 0.1 dB of the reference at equal steps").

Per seed the file stores the held-out PSNR at every checkpoint, the training loss of every step and an fp64 digest of the
initial weights (so the GPU side can prove it started from the same point); nothing of the reference's text travels.
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import nerf_oracle as O  # noqa: E402
from oracle import ref_shim  # noqa: E402
from oracle.make_golden import build_reference  # noqa: E402
from oracle.scenes import brick_scene  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "reference_psnr_curves.json")
B, S, N = 256, 64, 64
N_TRAIN_RAYS, N_VAL_RAYS = 40000, 4096
DRAW_ORDER = ("perturb_rand", "noise_coarse", "u", "noise_fine")


def init_digest(state_dicts):
    """fp64 (sum, sum of squares) over all tensors of the two default-initialised models, coarse then fine."""
    s = sum(v.double().sum().item() for sd in state_dicts for v in sd.values())
    q = sum((v.double() ** 2).sum().item() for sd in state_dicts for v in sd.values())
    return [s, q]


def batch_indices(perm, step, n_rays):
    return perm[((step - 1) * B) % (n_rays - B):][:B]


DEAD_AT, DEAD_BELOW_DB = 100, 9.0     # a run still at the all-white solution's PSNR (7.7 dB on this scene) at step 100 stays there (the dead-ReLU
                                       # density head every NeRF implementation knows): recorded as dead and cut short — no signal in it


def train_reference(seed, steps, checks, data, log=print):
    nerf, rend, _calls, replay = build_reference()
    losses_mod = ref_shim._load("_ref_losses", "losses.py")
    rays, rgbs, rays_val, rgb_val = data
    torch.manual_seed(seed)
    models = [nerf.NeRF(), nerf.NeRF()]                                    # coarse then fine (train.py:38-42)
    digest = init_digest([m.state_dict() for m in models])
    emb = [nerf.Embedding(3, 10), nerf.Embedding(3, 4)]
    loss_fn = losses_mod.MSELoss()
    opt = torch.optim.Adam([p for m in models for p in m.parameters()], lr=5e-4, eps=1e-8, weight_decay=0)
    perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(1000 + seed))
    curve, losses, t0 = {}, [], time.time()
    for step in range(1, steps + 1):
        idx = batch_indices(perm, step, rays.shape[0])
        rng = O.draw_rng(7000 * seed + step, B, S, N, 1.0)
        for key in DRAW_ORDER:
            replay.push("rand" if key in ("perturb_rand", "u") else "randn", rng[key])
        res = rend.render_rays(models, emb, rays[idx], S, False, 1.0, 0.0, N, 1024 * 32, True)   # train.py:49-71
        assert not replay.queue
        loss = loss_fn(res, rgbs[idx])
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if step in checks:
            with torch.no_grad():
                # validation form (train.py:119-138 -> forward -> render_rays with the training flags but perturb/noise drawn):
                # evaluated deterministically (perturb 0, noise 0) so the held-out PSNR is a function of the weights alone
                for key in ("noise_coarse", "noise_fine"):
                    replay.push("randn", torch.zeros(N_VAL_RAYS, S if key == "noise_coarse" else S + N))
                img = rend.render_rays(models, emb, rays_val, S, False, 0, 0.0, N, 1024 * 32, True)["rgb_fine"]
                assert not replay.queue
            curve[step] = O.psnr(img, rgb_val).item()
            log("seed %d step %d: loss %.5f  held-out PSNR %.3f dB  (%.0f s)" % (seed, step, losses[-1], curve[step], time.time() - t0))
            if step == DEAD_AT and curve[step] < DEAD_BELOW_DB:
                log("seed %d: dead init (%.2f dB at step %d), cut short" % (seed, curve[step], step))
                return {"seed": seed, "init_digest": digest, "dead": True, "psnr": curve, "loss": losses}
    return {"seed": seed, "init_digest": digest, "dead": False, "psnr": curve, "loss": losses}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--from-step", type=int, default=50)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--merge", nargs="*", default=[], help="other output files of this script (same settings) whose runs are merged into --out")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    data = brick_scene(N_TRAIN_RAYS, 1, "cpu") + brick_scene(N_VAL_RAYS, 2, "cpu")
    checks = set(range(a.from_step, a.steps + 1, a.every)) | {DEAD_AT}
    doc = {"what": "held-out PSNR@step of the REAL reference (unmodified models/nerf.py, models/rendering.py, losses.py; "
                   "torch-CPU fp32 autograd + torch.optim.Adam lr 5e-4 eps 1e-8) trained by oracle/make_psnr_curves.py",
           "torch": torch.__version__, "B": B, "S": S, "N": N, "steps": a.steps, "n_train_rays": N_TRAIN_RAYS, "n_val_rays": N_VAL_RAYS,
           "batches": "perm = randperm(n_train_rays, Generator(1000 + seed)); idx = perm[((step-1)*B) % (n_train_rays-B):][:B]",
           "draws": "oracle.nerf_oracle.draw_rng(7000 * seed + step, B, S, N, 1.0)", "runs": []}
    if os.path.exists(a.out):
        with open(a.out) as fh:
            old = json.load(fh)
        if all(old.get(k) == doc[k] for k in ("B", "S", "N", "steps", "n_train_rays", "n_val_rays")):
            doc["runs"] = old["runs"]
    for other in a.merge:
        with open(other) as fh:
            o = json.load(fh)
        assert all(o.get(k) == doc[k] for k in ("B", "S", "N", "steps", "n_train_rays", "n_val_rays")), other
        doc["runs"] += [r for r in o["runs"] if r["seed"] not in {x["seed"] for x in doc["runs"]}]
        doc["runs"].sort(key=lambda r: r["seed"])
    have = {r["seed"] for r in doc["runs"]}
    for seed in range(a.first_seed, a.first_seed + a.seeds):
        if seed in have:
            continue
        run = train_reference(seed, a.steps, checks, data)
        run["psnr"] = {str(k): v for k, v in run["psnr"].items()}
        doc["runs"].append(run)
        doc["runs"].sort(key=lambda r: r["seed"])
        with open(a.out, "w") as fh:
            json.dump(doc, fh, indent=0)
    print("wrote", a.out, "seeds", [r["seed"] for r in doc["runs"]])


if __name__ == "__main__":
    main()
