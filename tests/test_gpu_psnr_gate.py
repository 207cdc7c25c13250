"""GPU: the north-star PSNR gate — "PSNR within 0.1 dB of the reference at equal steps" (BASELINE.json; the reference's
lego runs end at 30.65-31.39 dB, test.ipynb:132 / README.md:161) — for the reduced-precision MLP modes.

No dataset is available offline, so the scene is the procedural lego-like `oracle.scenes.brick_scene`: sharp-edged geometry
with a hard-edged checker texture plus a fine grain above the bandwidth of the positional encoding (unrepresentable detail:
what makes a real scene plateau); PSNR@step ends at ~31.3 dB in fp32 with the compressed recipe (the README recipe's Adam
5e-4 with gamma-0.5 step decays, README.md:75-83).  Ground-truth colours come from closed-form quadrature, i.e. from neither
the HIP path nor the oracle.  Measured (profiles/archive/r03_psnr_gate_brick.json, 16 live seeds): bf16 +0.064 +- 0.037 dB, bf16_f8
+0.011 +- 0.026 dB relative to fp32.  Without the grain the same geometry with a harsher texture is still climbing at 30.8 dB
when the recipe ends and a run PAIR differs by +-0.6 dB of trajectory chaos (two fp32 runs that differ in a summation order do
too): bf16 +0.02 +- 0.21, bf16_f8 -0.18 +- 0.16 dB over 12 seeds (profiles/archive/r03_psnr_gate_brick_no_grain.json) — no detectable
deficit there either, at five times the noise.

Method: >= 16 LIVE init/jitter seeds (a seed whose fp32 run never leaves the all-white solution — the dead-ReLU density head
every NeRF implementation knows, identical in all precisions — is replaced, by a criterion on the fp32 run only); per seed the
fp32-MFMA path (the 1e-4-parity configuration, which tracks the CPU oracle to <= 0.05 dB, profiles/archive/r02_psnr_vs_oracle.json),
bf16 and bf16_f8 start from the same weights and consume the same batches and RNG draws; the statistic is the PAIRED difference
of the mean PSNR over the post-decay checkpoints.  Asserted: |mean difference| <= 0.1 dB with a standard error <= 0.05 dB.
"""
import statistics
from argparse import Namespace

import pytest
import torch

from oracle.scenes import BRICK_DEFAULT, brick_scene

pytestmark = pytest.mark.gpu

B, S, N = 1024, 64, 64
STEPS = 1200
LR_AT = {1: 5e-4, 600: 2.5e-4, 900: 1.25e-4}        # steplr, decay_gamma 0.5 (README.md:75-83), compressed
EVAL_AT = (1000, 1050, 1100, 1150, 1200)
N_LIVE, MAX_SEEDS = 16, 40
DEAD_BELOW_DB = 12.0                                # fp32 PSNR under this = the all-white dead init (7.3 dB on this scene)
DEAD_CHECK_AT = 150                                 # ... checked once, early in the fp32 run
N_TRAIN_RAYS, N_VAL_RAYS = 400000, 16384


def make_data(dev, **scene):
    rays, rgbs = brick_scene(N_TRAIN_RAYS, 1, dev, **scene)
    rays_val, rgb_val = brick_scene(N_VAL_RAYS, 2, dev, **scene)
    return rays, rgbs, rays_val, rgb_val


def train_curve(dtype, dev, data, init, jitter_seed, steps=STEPS, lr_at=LR_AT, eval_at=EVAL_AT, dead_check=None, N=N):
    """One training run of `steps` 1024-ray steps; returns {step: PSNR on the held-out rays}, or None when `dead_check` =
    (step, dB) finds the run still under `dB` at `step`."""
    from nerf_pl_amd.inference import batched_inference
    from nerf_pl_amd.system import NeRFSystem
    rays, rgbs, rays_val, rgb_val = data
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
    torch.manual_seed(jitter_seed)                           # the perturb / u draws (rendering.py:203, :39)
    perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(jitter_seed)).to(dev)
    curve = {}
    for step in range(1, steps + 1):
        if step in lr_at:
            for grp in opt.param_groups:
                grp["lr"] = lr_at[step]
        idx = perm[((step - 1) * B) % (rays.shape[0] - B):][:B]
        out = system.training_step({"rays": rays[idx], "rgbs": rgbs[idx]}, step)
        opt.zero_grad(set_to_none=True)
        out["loss"].backward()
        opt.step()
        if step in eval_at or (dead_check is not None and step == dead_check[0]):
            with torch.no_grad():
                img = batched_inference(system.models, system.embeddings, rays_val, S, N, False, 32768, True)["rgb_fine"]
            psnr = (-10 * torch.log10(torch.mean((img - rgb_val) ** 2))).item()
            if step in eval_at:
                curve[step] = psnr
            if dead_check is not None and step == dead_check[0] and psnr < dead_check[1]:
                return None
    return curve


def paired_statistics(dev, dtypes=("bf16", "bf16_f8"), n_live=N_LIVE, max_seeds=MAX_SEEDS, scene=None, log=print, **train_kw):
    """Runs fp32 + `dtypes` per seed until `n_live` live seeds exist.  Returns a JSON-able summary."""
    from nerf_pl_amd.models import NeRF
    data = make_data(dev, **(scene or {}))
    eval_at = train_kw.get("eval_at", EVAL_AT)
    finals = {dt: [] for dt in ("fp32",) + tuple(dtypes)}
    curves, seeds, dead = [], [], []
    for seed in range(max_seeds):
        if len(seeds) >= n_live:
            break
        torch.manual_seed(seed)
        init = [NeRF().state_dict(), NeRF().state_dict()]    # default nn.Linear init, coarse then fine (train.py:38-42)
        c32 = train_curve("fp32", dev, data, init, 1000 + seed, dead_check=(DEAD_CHECK_AT, DEAD_BELOW_DB), **train_kw)
        if c32 is None:
            dead.append(seed)
            log("seed %d: dead init (fp32 under %.0f dB at step %d), replaced" % (seed, DEAD_BELOW_DB, DEAD_CHECK_AT))
            continue
        f32 = sum(c32[s] for s in eval_at) / len(eval_at)
        row = {"seed": seed, "fp32": c32}
        finals["fp32"].append(f32)
        for dt in dtypes:
            c = train_curve(dt, dev, data, init, 1000 + seed, **train_kw)
            row[dt] = c
            finals[dt].append(sum(c[s] for s in eval_at) / len(eval_at))
        seeds.append(seed)
        curves.append(row)
        log("seed %d: " % seed + "  ".join("%s %.3f" % (dt, finals[dt][-1]) for dt in finals))
    out = {"scene": dict(BRICK_DEFAULT, **(scene or {})), "seeds": seeds, "dead_seeds": dead, "eval_at": list(eval_at),
           "mean_psnr": {dt: round(statistics.mean(v), 4) for dt, v in finals.items() if v},
           "per_seed": {dt: [round(x, 4) for x in v] for dt, v in finals.items()}, "paired": {}, "curves": curves}
    for dt in dtypes:
        d = [a - b for a, b in zip(finals[dt], finals["fp32"])]
        if len(d) > 1:
            out["paired"][dt] = {"mean": round(statistics.mean(d), 4), "stderr": round(statistics.stdev(d) / len(d) ** 0.5, 4),
                                 "stdev": round(statistics.stdev(d), 4), "per_seed": [round(x, 4) for x in d]}
    return out


def test_psnr_within_0p1_db_of_fp32_at_equal_steps(dev):
    res = paired_statistics(dev)
    print("PSNR gate:", {k: res[k] for k in ("scene", "seeds", "dead_seeds", "mean_psnr", "paired")})
    assert len(res["seeds"]) >= N_LIVE, res["dead_seeds"]
    # the scene is in the regime the metric is quoted in (lego: 30.65-31.39 dB)
    assert 29.5 <= res["mean_psnr"]["fp32"] <= 33.0, res["mean_psnr"]
    for dt, p in res["paired"].items():
        assert p["stderr"] <= 0.05, (dt, p)
        assert abs(p["mean"]) <= 0.1, (dt, p)


def test_psnr_gate_at_the_headline_sampling_64_plus_128(dev):
    """The same paired gate at the HEADLINE sampling of BASELINE configs[2] — 64 coarse + 128 importance samples, what bench.py
    times — on 8 live seeds (the 16-seed gate above runs the README recipe's 64 + 64): |mean paired difference| <= 0.1 dB,
    standard error <= 0.07 dB."""
    res = paired_statistics(dev, n_live=8, N=128)
    print("PSNR gate 64+128:", {k: res[k] for k in ("seeds", "dead_seeds", "mean_psnr", "paired")})
    assert len(res["seeds"]) >= 8, res["dead_seeds"]
    assert 29.5 <= res["mean_psnr"]["fp32"] <= 33.5, res["mean_psnr"]
    for dt, p in res["paired"].items():
        assert p["stderr"] <= 0.07, (dt, p)
        assert abs(p["mean"]) <= 0.1, (dt, p)


def test_fp32_comparator_tracks_the_oracle_on_the_gate_scene(dev):
    """The gate above compares bf16 / bf16_f8 with the HIP fp32-MFMA path; this leg ties that comparator to the CPU oracle (the
    restatement pinned to the real reference by tests/test_oracle_*.py) ON THE GATE'S OWN SCENE: the same default init, the same
    256-ray batches of brick_scene and the same replayed draws through 150 Adam steps of the oracle (torch-CPU autograd +
    torch.optim.Adam) and of the HIP path (the fused training node + FlatAdam, what bench.py times), PSNR on 4,096 held-out rays
    at steps 120 / 130 / 140 / 150: the end of the window within 0.10 dB, the mean of the four within 0.12 dB, while the run climbs from ~21 to ~22.4 dB.  (Longer windows are not
    comparable run-to-run: two fp32 runs that differ in one summation order drift apart by trajectory chaos alone.)"""
    import os
    from oracle import nerf_oracle as O
    from nerf_pl_amd.inference import batched_inference
    from nerf_pl_amd.models import NeRF
    from nerf_pl_amd.models.train_step import render_rays_train
    from nerf_pl_amd.system import NeRFSystem
    Bo, steps, seed = 256, 150, 0
    checks = (120, 130, 140, 150)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))      # torch-CPU oversubscribes badly on many-core hosts (7x slower at 128)
    rays, rgbs = brick_scene(40000, 1, "cpu")
    rays_val, rgb_val = brick_scene(4096, 2, "cpu")
    torch.manual_seed(seed)
    init = [NeRF().state_dict(), NeRF().state_dict()]
    perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(1000 + seed))

    def batch(step):
        idx = perm[((step - 1) * Bo) % (rays.shape[0] - Bo):][:Bo]
        return rays[idx], rgbs[idx], O.draw_rng(7000 * seed + step, Bo, S, N, 1.0)

    # ---- oracle ----
    params = [{k: v.clone().requires_grad_(True) for k, v in sd.items()} for sd in init]
    opt = torch.optim.Adam([v for p in params for v in p.values()], lr=5e-4, eps=1e-8)
    want = {}
    for step in range(1, steps + 1):
        r, t, rng = batch(step)
        loss = O.mse_loss(O.render_rays(params, r, S, False, 1.0, 0.0, N, True, False, rng=rng), t)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step in checks:
            with torch.no_grad():
                img = O.render_rays(params, rays_val, S, False, 0, 0.0, N, True, False)["rgb_fine"]
            want[step] = O.psnr(img, rgb_val).item()
    torch.set_num_threads(threads)
    # ---- HIP fp32: the fused training node on the same draws ----
    hp = Namespace(N_samples=S, N_importance=N, use_disp=False, perturb=1.0, noise_std=0.0, chunk=32768, loss_type="mse", lr=5e-4,
                   weight_decay=0, decay_step=[10 ** 9], decay_gamma=0.5, white_back=True, optimizer="adam", lr_scheduler="steplr")
    system = NeRFSystem(hp)
    system.nerf_coarse.load_state_dict(init[0])
    system.nerf_fine.load_state_dict(init[1])
    for m in system.models:
        m.mlp_dtype = "fp32"
    system = system.to(dev)
    (hopt,), _ = system.configure_optimizers()
    got = {}
    for step in range(1, steps + 1):
        r, t, rng = batch(step)
        draws = {k: v.to(dev) for k, v in rng.items() if k in ("perturb_rand", "u")}
        _, loss, _ = render_rays_train(system.models, system.embeddings, r.to(dev), t.to(dev), S, False, 1.0, 0.0, N, True, draws=draws)
        hopt.zero_grad(set_to_none=True)
        loss.backward()
        hopt.step()
        if step in checks:
            with torch.no_grad():
                img = batched_inference(system.models, system.embeddings, rays_val.to(dev), S, N, False, 32768, True)["rgb_fine"]
            got[step] = (-10 * torch.log10(torch.mean((img.cpu() - rgb_val) ** 2))).item()
    diffs = [got[s] - want[s] for s in checks]
    print("fp32 HIP vs oracle on brick_scene, PSNR (HIP, oracle) at steps %s:" % (checks,), {s: (round(got[s], 3), round(want[s], 3)) for s in checks},
          "mean difference %.3f dB" % (sum(diffs) / len(diffs)))
    assert want[150] - want[120] > 0.5 and want[150] > 20.0      # a live, climbing run (not a dead init)
    # Between steps 50 and 125 this run climbs up to 0.13 dB PER STEP (14.2 -> 18.1 -> 21.4 dB at steps 50 / 100 / 125), so a single
    # checkpoint there carries the phase noise of the trajectory (measured HIP - oracle: +0.001, -0.099, -0.028, -0.029 dB at steps
    # 50 / 100 / 125 / 150): the statistic is taken where the curve has flattened — mean over the last checkpoints and the end
    # Measured (round 5, same box, `tools/gpu_calls/r05_08.sh`): -0.110 / -0.075 / -0.074 / -0.072 dB with sample_pdf's row total in
    # ATen's order (the default), -0.089 / -0.060 / -0.051 / -0.029 dB with the correctly rounded total — the two differ ONLY in which
    # 0.1-0.5 % of the fine samples sit on the other side of a bin edge (last-bit knife edges), i.e. 0.04 dB at step 150 is what 150
    # steps of trajectory make of a last-bit choice; the single-launch forward and the four launches give identical numbers
    # (bit-identical steps).  The bounds sit at ~2x that amplitude.
    assert abs(diffs[-1]) <= 0.10, (got, want)                       # the end of the window
    assert abs(sum(diffs) / len(diffs)) <= 0.12, (got, want)         # the last four checkpoints (the HIP run is ~1 step behind the
    assert max(abs(d) for d in diffs) <= 0.15, (got, want)           # oracle's on this seed, closing)
