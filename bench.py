"""bench.py — rays/sec of the NeRF hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32] [--mode train|render]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 1024 synthetic rays
per GPU, 64 coarse + 128 importance samples, NeRF D=8 W=256 coarse + fine, perturb=1, noise_std=0,
white background (the README lego recipe).  Inputs are resident in HBM before the timed region.

A step (default `--mode train`) is ONE FULL TRAINING STEP of the reference's NeRFSystem.training_step
contract: render_rays forward (coarse + fine, 256 MLP evaluations per ray) -> MSE loss + PSNR ->
backward through compositing and both MLPs -> [N>1: RCCL all-reduce of the gradients] -> Adam step.
`--mode render` times the forward `render_rays` alone.  Ray batches shard across ranks (weak scaling:
1024 rays per GPU, like the reference's DDP).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_POINT_FULL = 1186816      # 593,408 MAC, GEMMs only, no padding counted
FLOP_PER_POINT_BWD = 2302208       # dX chain 557,696 MAC (no dX into the encodings) + dW 593,408 MAC
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
PEAK_TFLOPS = {"bf16": 2500.0, "bf16_f8": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md dense MFMA peaks


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "bf16_f8", "fp32"])
    ap.add_argument("--mode", default="train", choices=["train", "render", "eval"])
    ap.add_argument("--image-rays", type=int, default=640000, help="--mode eval: rays per image (800x800), sharded over ranks")
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--n-samples", type=int, default=64)
    ap.add_argument("--n-importance", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue the training step eagerly instead of replaying a hipGraph")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


# ---- seeded synthetic inputs (BASELINE.json: no dataset / checkpoint offline).  Self-contained on purpose: the only
# part of this file that touches the CPU checker under its test-infrastructure directory is cpu_baseline(). ----
PARAM_SHAPES = [("xyz_encoding_%d.0" % (i + 1), 256, 63 if i == 0 else (319 if i == 4 else 256)) for i in range(8)] + \
               [("xyz_encoding_final", 256, 256), ("dir_encoding.0", 128, 283), ("sigma", 1, 256), ("rgb.0", 3, 128)]


def synth_params(seed, sigma_gain=1.0, sigma_bias=0.0):
    """nn.Linear's default U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from numpy PCG64 (identical on every rank and box);
    the density head is rescaled so that opacity is non-trivial (a trained-like field)."""
    import math

    import numpy as np
    rng = np.random.default_rng(seed)
    p = {}
    for name, fo, fi in PARAM_SHAPES:
        b = 1.0 / math.sqrt(fi)
        p[name + ".weight"] = torch.from_numpy(rng.uniform(-b, b, size=(fo, fi)).astype(np.float32))
        p[name + ".bias"] = torch.from_numpy(rng.uniform(-b, b, size=(fo,)).astype(np.float32))
    p["sigma.weight"] = p["sigma.weight"] * sigma_gain
    p["sigma.bias"] = p["sigma.bias"] * sigma_gain + sigma_bias
    return p


def synth_rays(seed, n):
    """Blender-style rays (n,8): origins (0,0,4)+0.1N, unit directions aimed near the scene centre, near 2, far 6
    (blender.py:34-35)."""
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(n, 3, generator=g)
    d = 0.8 * torch.randn(n, 3, generator=g) - o
    d = d / d.norm(dim=-1, keepdim=True)
    return torch.cat([o, d, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0)], 1).float().contiguous()


def cpu_baseline(B, S, N, seconds, train):
    """The pinned CPU oracle (torch-CPU restatement of the reference's render_rays, kind='port') timed
    on this node's host cores on the same workload shape; bounded to ~`seconds` of CPU work."""
    from oracle import nerf_oracle as O
    params = [O.make_params(0), O.make_params(1)]
    rays = O.make_rays(0, B, "blender")
    tgt = torch.rand(B, 3, generator=torch.Generator().manual_seed(0))
    rng = O.draw_rng(0, B, S, N, 1.0)
    if train:
        for d in params:
            for v in d.values():
                v.requires_grad_(True)
        opt = torch.optim.Adam([v for d in params for v in d.values()], lr=5e-4)

    def one(rays_, tgt_, rng_):
        if not train:
            with torch.no_grad():
                O.render_rays(params, rays_, S, False, 1.0, 0, N, True, False, rng=rng_)
            return
        res = O.render_rays(params, rays_, S, False, 1.0, 0, N, True, False, rng=rng_)
        loss = O.mse_loss(res, tgt_)
        opt.zero_grad()
        loss.backward()
        opt.step()

    ncpu = os.cpu_count() or 1
    # torch-CPU oversubscribes badly on many-core hosts: pick the fastest of a few thread counts on a
    # 1/8-size probe, then time the full workload with it.
    best, best_t = 1, float("inf")
    Bp = max(32, B // 8)
    sub = {k: v[:Bp] for k, v in rng.items()}
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nt)
        one(rays[:Bp], tgt[:Bp], sub)
        t0 = time.perf_counter()
        one(rays[:Bp], tgt[:Bp], sub)
        t = time.perf_counter() - t0
        if t < best_t:
            best, best_t = nt, t
    torch.set_num_threads(best)
    one(rays, tgt, rng)  # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        one(rays, tgt, rng)
        reps += 1
        if time.perf_counter() - t0 > seconds or reps >= 50:
            break
    dt = time.perf_counter() - t0
    what = "training step (fwd+loss+bwd+Adam)" if train else "render_rays fwd"
    return {"value": round(B * reps / dt, 1), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d reps of the oracle's %s on %d rays x (%d+%d), torch-CPU fp32, %.1f s"
                      % (reps, what, B, S, N, dt)}


def main():
    a = parse()
    # stdout carries exactly ONE JSON line: libraries that print banners to fd 1 (RCCL prints its version block on the
    # first communicator) are diverted to stderr for the whole run; the result goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from nerf_pl_amd import ops
    from nerf_pl_amd.models import rendering
    from nerf_pl_amd.parallel import GradSync
    from nerf_pl_amd.system import GraphedTrainStep, NeRFSystem

    B, S, N = a.rays, a.n_samples, a.n_importance
    hp = Namespace(N_samples=S, N_importance=N, use_disp=False, perturb=1.0, noise_std=0.0, chunk=1024 * 32,
                   loss_type="mse", lr=5e-4, weight_decay=0, decay_step=[2, 4, 8], decay_gamma=0.5, white_back=True)
    system = NeRFSystem(hp)
    # random-init weights of the named architecture (identical on every rank = DDP replicas), density head
    # scaled so that opacity is non-trivial
    system.nerf_coarse.load_state_dict(synth_params(100, 4.0, 0.2))
    system.nerf_fine.load_state_dict(synth_params(101, 4.0, 0.2))
    for m in system.models:
        m.mlp_dtype = a.dtype
    system = system.to(dev)
    models, emb = system.models, system.embeddings
    (opt,), _ = system.configure_optimizers()
    grad_sync = GradSync(models) if world > 1 else None
    rays = synth_rays(1234 + rank, B).to(dev)                 # each rank draws its own batch
    rgbs = torch.rand(B, 3, generator=torch.Generator().manual_seed(rank)).to(dev)
    batch = {"rays": rays, "rgbs": rgbs}
    torch.manual_seed(rank)

    def render_step():
        with torch.no_grad():
            return rendering.render_rays(models, emb, rays, S, False, 1.0, 0.0, N, 1024 * 32, True)

    # The training step (fwd, loss, bwd, [all-reduce], Adam: ~45 launches) is replayed as ONE hipGraph after 3 eager
    # steps; same work per step, ~15 us of host time instead of ~1.5 ms.  Falls back to eager issue if capture fails.
    state = {"graphed": GraphedTrainStep(system, opt, grad_sync, warmup=3) if (a.mode == "train" and not a.no_graph) else None}

    def eager_step():
        out = system.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        out["loss"].backward()
        if grad_sync is not None:
            grad_sync.sync()
        opt.step()
        return out

    def train_step():
        g = state["graphed"]
        if g is None:
            return eager_step()
        try:
            return g(batch)
        except Exception as e:  # noqa: BLE001 - capture problems must not kill the benchmark
            print("[bench] hipGraph capture failed (%s: %s); continuing eagerly" % (type(e).__name__, e), file=sys.stderr, flush=True)
            state["graphed"] = None
            torch.cuda.synchronize()
            return eager_step()

    # --mode eval (BASELINE.json configs[4]): one step = one full 800x800 image through eval.py's batched_inference
    # contract (32768-ray chunks, test_time: sigma-only coarse pass), each chunk a hipGraph replay; the ray list is
    # sharded contiguously over the ranks (strong scaling) and the finished pixels are all-gathered.
    eval_state = {}
    if a.mode == "eval":
        from nerf_pl_amd.inference import GraphRenderer
        from nerf_pl_amd.parallel import render_sharded
        eval_state["rays"] = synth_rays(77, a.image_rays).to(dev)
        eval_state["gr"] = GraphRenderer(models, emb, S, N, False, True)

    def eval_step():
        return render_sharded(eval_state["gr"], eval_state["rays"], keys=("rgb_fine", "depth_fine"))

    step = train_step if a.mode == "train" else (eval_step if a.mode == "eval" else render_step)
    for _ in range(max(a.warmup, 5) if a.mode == "train" else a.warmup):     # >= 5: 3 eager + capture + 1 replay
        step()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        extra = {}
        if a.mode == "train":                                  # forward-only rate of the same workload
            for _ in range(5):
                render_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(30):
                render_step()
            torch.cuda.synchronize()
            extra["render_fwd_rays_per_s_per_gpu"] = round(B * 30 / (time.perf_counter() - t1), 1)

        # ---- dominant kernel: fine-pass fused MLP forward (B x (S+N) points); HIP events on the launch
        # stream (kernels are launched on torch's current stream, which is what torch.cuda.Event times) ----
        with torch.no_grad():
            z = ops.sample_coarse_z(rays, S, False, 0.0)
            zf = ops.fine_z(z, torch.rand(B, S, device=dev), N)
            pk = models[1].packed_weights()                     # pack once: the events bracket the MLP kernel alone
            for _ in range(5):
                ops.mlp_fwd_rays(rays, zf, pk, False, a.dtype)
            reps = 30
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            ev[0].record()
            for i in range(reps):
                ops.mlp_fwd_rays(rays, zf, pk, False, a.dtype)
                ev[i + 1].record()
            torch.cuda.synchronize()
            ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
            avg_ms = sum(ms) / len(ms)
        flops = FLOP_PER_POINT_FULL * B * (S + N)
        ach = flops / (avg_ms * 1e-3) / 1e12
        kname = "mlp_fwd_kernel<%s,rays,full> %dx%d points" % (a.dtype, B, S + N)
        traffic = None      # HBM bytes per launch from the PMC passes of the same command (profiles/), else null
        if os.path.exists(TRAFFIC_JSON):
            with open(TRAFFIC_JSON) as fh:
                traffic = json.load(fh).get("mlp_fwd_%s_%dx%d" % (a.dtype, B, S + N), {}).get("hbm_bytes_per_launch")
        roof = {"bound": "mfma", "kernel": kname,
                "achieved": round(ach, 2), "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                "frac": round(ach / PEAK_TFLOPS[a.dtype], 4), "traffic": traffic,
                "avg_launch_us": round(avg_ms * 1e3, 2), "min_launch_us": round(ms[0] * 1e3, 2)}
        extra["roofline"] = roof
        if a.mode == "train":
            # backward of the same fine pass: fwd(save) once, then nerfhip_mlp_bwd = chain + dW + reduce kernels
            acts = ops.alloc_acts(B * (S + N), a.dtype, dev)
            out_f = ops.mlp_fwd_rays(rays, zf, models[1].packed_weights(), False, a.dtype, save=acts)
            g_out = torch.randn_like(out_f)
            pb = models[1].packed_weights_bwd()
            for _ in range(3):
                ops.mlp_bwd(g_out, out_f, pb, acts, a.dtype)
            evb = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            evb[0].record()
            for i in range(10):
                ops.mlp_bwd(g_out, out_f, pb, acts, a.dtype)
                evb[i + 1].record()
            torch.cuda.synchronize()
            bms = sum(evb[i].elapsed_time(evb[i + 1]) for i in range(10)) / 10
            bach = FLOP_PER_POINT_BWD * B * (S + N) / (bms * 1e-3) / 1e12
            extra["roofline_bwd"] = {"bound": "hbm+mfma", "kernel": "nerfhip_mlp_bwd<%s> = bwd_chain + bwd_dw + reduce, %dx%d points"
                                     % (a.dtype, B, S + N), "achieved": round(bach, 2), "peak": PEAK_TFLOPS[a.dtype],
                                     "unit": "TFLOP/s", "frac": round(bach / PEAK_TFLOPS[a.dtype], 4),
                                     "avg_launch_us": round(bms * 1e3, 2)}
            del acts, out_f, g_out

        total_rays = (a.image_rays if a.mode == "eval" else world * B) * a.steps
        out = {
            "metric": ("rays/sec (64+128 samples), full training step: render_rays fwd + MSE + bwd + grad all-reduce + Adam"
                       if a.mode == "train" else
                       "rays/sec (64+128 samples), full-image inference (eval.py batched_inference, test_time, 32768-ray hipGraph chunks)"
                       if a.mode == "eval" else
                       "rays/sec (64+128 samples), render_rays forward only (train-mode: coarse+fine rgb)"),
            "value": round(total_rays / dt, 1), "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if a.mode == "eval" else "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ("configs[4]: %d-ray image in 32768-ray chunks x (%d+%d) samples, test_time, NeRF D8 W256 "
                                    "coarse(sigma-only)+fine, %s MFMA MLP, mode=eval" % (a.image_rays, S, N, a.dtype))
                                   if a.mode == "eval" else
                                   ("configs[2]: %d rays/GPU x (%d+%d) samples, NeRF D8 W256 coarse+fine, perturb=1 "
                                    "noise_std=0 white_back, %s MFMA MLP, mode=%s" % (B, S, N, a.dtype, a.mode)),
                       "rays_per_gpu": (a.image_rays // world if a.mode == "eval" else B), "N_samples": S, "N_importance": N,
                       "issue": ("hipGraph replay of the whole step" if (a.mode == "train" and state["graphed"] is not None
                                                                         and state["graphed"].graph is not None)
                                 else "hipGraph replay per 32768-ray chunk" if a.mode == "eval" else "eager"),
                       "parallelism": "ray-sharded x%d%s" % (world, ", RCCL grad all-reduce" if world > 1 and a.mode == "train" else "")},
        }
        out.update(extra)
        if not a.no_cpu_baseline and world == 1:                # reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(B, S, N, a.cpu_seconds, a.mode == "train")
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
