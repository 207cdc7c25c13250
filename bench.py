"""bench.py — rays/sec of the NeRF hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|bf16_f8|fp32] [--mode train|render|eval]
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
FLOP_PER_POINT_DX = 1115392        # backward chain: 557,696 MAC (no dX into the encodings)
FLOP_PER_POINT_DW = 1186816        # weight-gradient GEMM: 593,408 MAC
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
PEAK_TFLOPS = {"bf16": 2500.0, "bf16_f8": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md dense MFMA peaks of the forward / dX chain
PEAK_TFLOPS_FP8 = 5000.0           # ... and of the MX-scaled fp8 MFMA the dW GEMM of bf16_f8 runs on
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy)
FLOP_PER_RAY_EVAL = 64 * 982528 + 192 * 1186816      # test_time render: sigma-only coarse pass + full fine pass (290.7 M)
# `dtype` of the JSON line = the NARROWEST arithmetic inside the timed region
DTYPE_LABEL = {"bf16_f8": "bf16+fp8(dW)", "bf16": "bf16", "fp32": "f32"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "bf16_f8", "fp32"],
                    help="bf16 (default; the arithmetic BASELINE.json configs[2] names): bf16 MFMA everywhere, saved tensors in bf16; "
                         "bf16_f8: the same forward + dX chain, saved activations/dY stored as block-scaled 8-bit floats and the dW GEMM on "
                         "the MX-scaled fp8 MFMA (reported beside the headline as f8_dw_ms_per_step); fp32: exact-fp32 MFMA (parity)")
    ap.add_argument("--mode", default="train", choices=["train", "render", "eval"])
    ap.add_argument("--image-rays", type=int, default=640000, help="--mode eval: rays per image (800x800), sharded over ranks")
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--n-samples", type=int, default=64)
    ap.add_argument("--n-importance", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue the training step eagerly instead of replaying a hipGraph")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--fixed-batch", action="store_true", help="replay one resident batch instead of drawing a fresh one per step")
    ap.add_argument("--modular-step", action="store_true",
                    help="training_step through the modular autograd graph (render_rays -> MSELoss) instead of the fused node (A/B)")
    ap.add_argument("--fuse-adam", action="store_true",
                    help="N=1: apply Adam inside the dW-reduce kernel instead of as its own launch (one launch fewer; measured 0-15 us "
                         "SLOWER per step in six same-call pairs on four boxes, so off by default)")
    ap.add_argument("--no-fuse-adam", action="store_true", help="(the default since round 3; accepted for older command lines)")
    ap.add_argument("--no-extras", action="store_true",
                    help="train mode: skip the side measurements (eval / render / fp8-dW variant / configs[1] and configs[3] steps / PMC passes)")
    ap.add_argument("--settle", type=int, default=None,
                    help="untimed replays between the capture and the W warm-up steps (device clocks; default by mode, 0 = none)")
    ap.add_argument("--series", action="store_true", help="diagnosis: per-step durations of the timed steps (events between the steps) on stderr")
    ap.add_argument("--no-pmc", action="store_true", help="do not take HBM counters in-run (rocprofv3 --pmc passes of this build's kernels)")
    ap.add_argument("--pmc-launch", action="store_true", help=argparse.SUPPRESS)   # child mode of the PMC passes: launch the kernels, print nothing
    ap.add_argument("--force-dist", action="store_true",
                    help="N=1: initialise the RCCL process group and route gradients through GradSync anyway (A/B of the N>1 step)")
    ap.add_argument("--sync-in-graph", type=int, default=None, choices=[0, 1],
                    help="N>1 graphed step: 1 = ONE graph with the grad-ready-hook all-reduces inside (overlapped), 0 = two graphs "
                         "with the collectives issued eagerly in between (default: GraphedTrainStep's)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-exec under torch.distributed.run, one rank per GPU, and pass
    its exit status on.  A world that does not match --gpus never prints a line."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != a.gpus:
            print("[bench] WORLD_SIZE=%s but --gpus %d: refusing to report a line for a different world size" % (env_world, a.gpus),
                  file=sys.stderr, flush=True)
            sys.exit(2)
        return
    if a.gpus <= 1:
        return
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus:
        print("[bench] --gpus %d but only %d GPU(s) visible: not running (no silent n_gpus=1 line)" % (a.gpus, n_dev),
              file=sys.stderr, flush=True)
        sys.exit(3)
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching %d ranks: %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd))


def init_world(a, backend="nccl"):
    """The launch contract's rank logic: read RANK / LOCAL_RANK / WORLD_SIZE, bring up the process group (backend "nccl" IS RCCL
    on ROCm; "gloo" lets tests/test_distributed_cpu.py drive this very function with two CPU processes), refuse a world that is
    not --gpus, and ask the communicator itself how many ranks it has (every rank adds a 1 to a sum all-reduce).
    Returns (dist | None, world, rank, local_rank, communicator ranks | None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    nranks = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            where = torch.device("cuda", local)
        else:
            dist.init_process_group(backend)
            where = torch.device("cpu")
        if dist.get_world_size() != a.gpus:
            print("[bench] RCCL world size %d != --gpus %d" % (dist.get_world_size(), a.gpus), file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(2)
        # what RCCL itself thinks the communicator is: every rank contributes a 1 to a sum all-reduce
        one = torch.ones(1, device=where)
        dist.all_reduce(one)
        nranks = int(one.item())
        if nranks != a.gpus:
            print("[bench] RCCL all-reduce saw %d ranks, --gpus %d" % (nranks, a.gpus), file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(2)
    return dist, world, rank, local, nranks


def timed_region(step_fn, warmup, steps, dist, device, series=None):
    """The contract's timing: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by a barrier + device synchronize on
    both sides; the MAX over the ranks is the job's time.  `series` (a list, diagnosis only: --series) receives the per-step
    durations in ms from events recorded between the steps on the current stream."""
    on_gpu = torch.device(device).type == "cuda"

    def sync():
        if on_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()
    for _ in range(warmup):
        step_fn()
    sync()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if (series is not None and on_gpu) else None
    t0 = time.perf_counter()
    if evs is None:
        for _ in range(steps):
            step_fn()
    else:
        evs[0].record()
        for i in range(steps):
            step_fn()
            evs[i + 1].record()
    sync()
    dt_ = time.perf_counter() - t0
    if evs is not None:
        series.extend(round(evs[i].elapsed_time(evs[i + 1]), 4) for i in range(steps))
    if dist is not None:
        t = torch.tensor([dt_], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_ = float(t.item())
    return dt_


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


def synth_store(seed, dev, n_img=20, hw=200):
    """Device-resident synthetic training set in the reference's Blender layout (blender.py:42-69): camera poses on a
    radius-4 sphere looking at the origin + random pixel colours; batches are drawn and their rays generated on the GPU
    (nerf_pl_amd.rays.RayStore), so a training batch never crosses PCIe."""
    from nerf_pl_amd.rays import RayStore
    g = torch.Generator().manual_seed(seed)
    c = torch.nn.functional.normalize(torch.randn(n_img, 3, generator=g), dim=-1) * 4.0
    fwd = torch.nn.functional.normalize(-c, dim=-1)                       # camera looks down its -z axis at the origin
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.nn.functional.normalize(torch.cross(fwd, up, dim=-1), dim=-1)
    upv = torch.cross(right, fwd, dim=-1)
    poses = torch.stack([right, upv, -fwd, c], -1).float().contiguous()   # (n_img, 3, 4) = [R | t]
    rgbs = torch.rand(n_img * hw * hw, 3, generator=g)
    return RayStore(poses.to(dev), rgbs.to(dev), hw, hw, 0.5 * hw / 0.3, 2.0, 6.0)


def synth_store_ndc(seed, dev, n_img=20, hw=200):
    """The same in the reference's forward-facing LLFF layout (llff.py:236-253): cameras near the origin looking down -z with
    small rotations, rays converted to NDC (near plane 1.0), bounds 0..1, non-unit directions."""
    from nerf_pl_amd.rays import RayStore
    g = torch.Generator().manual_seed(seed)
    w = 0.1 * torch.randn(n_img, 3, generator=g)                          # small axis-angle rotations
    K = torch.zeros(n_img, 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -w[:, 2], w[:, 1], w[:, 2], -w[:, 0], -w[:, 1], w[:, 0]
    R = torch.matrix_exp(K)
    t = 0.3 * torch.randn(n_img, 3, 1, generator=g)
    poses = torch.cat([R, t], -1).float().contiguous()
    rgbs = torch.rand(n_img * hw * hw, 3, generator=g)
    return RayStore(poses.to(dev), rgbs.to(dev), hw, hw, 0.5 * hw / 0.35, 0.0, 1.0, use_ndc=True, ndc_near_plane=1.0)


# ---- HBM counters of THIS run's kernels (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes with --kernel-trace only,
# as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled: gfx950 tallies 64 B per 128-B request of the wide reads these kernels
# make).  The passes re-run this file in `--pmc-launch` mode: every MLP kernel of the step twice on resident buffers. ----
def pmc_launch(a):
    from nerf_pl_amd import ops
    from nerf_pl_amd.models import NeRF
    dev = torch.device("cuda", 0)
    B, S, N = a.rays, a.n_samples, a.n_importance
    models = []
    for sd in (100, 101):
        m = NeRF()
        m.load_state_dict(synth_params(sd, 4.0, 0.2))
        m.mlp_dtype = a.dtype
        models.append(m.to(dev))
    rays = synth_rays(1234, B).to(dev)
    with torch.no_grad():
        z = ops.sample_coarse_z(rays, S, False, 0.0)
        zf = ops.fine_z(z, torch.rand(B, S, device=dev), N)
        pk = models[1].packed_weights(a.dtype)
        entries = []
        for model, zz in ((models[1], zf), (models[0], z)):
            acts = ops.alloc_acts(zz.numel(), a.dtype, dev)
            pf, pb = model.packed_weights_train(a.dtype)
            pf, pb = pf.clone(), pb.clone()
            for _ in range(2):
                raw = ops.mlp_fwd_rays(rays, zz, pf, False, a.dtype, save=acts)
            entries.append((torch.randn_like(raw), raw, pb, acts))
        ws = {}
        for _ in range(2):
            ops.mlp_bwd_multi(entries, a.dtype, workspace=ws)           # chain x 2, merged dW, merged reduce
            ops.mlp_fwd_rays(rays, zf, pk, False, a.dtype)               # the inference forward (north star)
        if ops.render_supported(B, S, N, a.dtype):                       # the step's forward as ONE launch
            tgt, pr, u = torch.rand(B, 3, device=dev), torch.rand(B, S, device=dev), torch.rand(B, N, device=dev)
            pk_c = models[0].packed_weights(a.dtype)
            for _ in range(2):
                ops.render_train_fwd(rays, tgt, 2.0 / (3 * B), S, N, pk_c, pk, a.dtype, entries[1][3], entries[0][3], False, 1.0, pr, None, None,
                                     0.0, True, u)
    torch.cuda.synchronize()


def pmc_collect(a, note):
    """{traffic key: {'hbm_bytes_per_launch': ...}} of this build's MLP kernels at this run's sizes, or {} (and why, in note)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        note["traffic_note"] = "rocprofv3 not on PATH: " + note.get("traffic_note", "")
        return {}
    B, S, N = a.rays, a.n_samples, a.n_importance
    P_f, P_c = B * (S + N), B * S
    grid = lambda P: (P + 255) // 256 * 512                              # bf16 kernels: 256 points = 8 waves per workgroup
    vals = {}
    tmp = tempfile.mkdtemp(prefix="nerfhip_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for C in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, C)
            cmd = ["rocprofv3", "--pmc", C, "--kernel-trace", "-f", "csv", "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-launch", "--dtype", a.dtype, "--rays", str(B), "--n-samples", str(S), "--n-importance", str(N)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=90)
            fs = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not fs:
                note["traffic_note"] = "in-run rocprofv3 --pmc %s pass failed (rc %s): %s" % (C, r.returncode, note.get("traffic_note", ""))
                return {}
            per = collections.defaultdict(float)
            csv.field_size_limit(1 << 30)
            for row in csv.DictReader(open(fs[0])):
                if row["Counter_Name"] == C and "mlp_" in row["Kernel_Name"]:
                    name = row["Kernel_Name"].replace("void ", "").replace("nerfhip::", "").split("(")[0]
                    per[(name, int(row["Grid_Size"]), row["Dispatch_Id"])] += float(row["Counter_Value"])
            agg = collections.defaultdict(list)
            for (name, g, _), v in per.items():
                agg[(name, g)].append(v)
            for k, v in agg.items():
                vals.setdefault(k, {})[C] = sum(v) / len(v)
    except Exception as e:  # noqa: BLE001 - counters are an extra, never fatal
        note["traffic_note"] = "in-run PMC passes failed (%s: %s): %s" % (type(e).__name__, e, note.get("traffic_note", ""))
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    db = {}
    for (name, g), cs in vals.items():
        if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
            continue
        base = name.split("<")[0]
        targs = name[name.index("<") + 1:name.rindex(">")].replace(" ", "").split(",") if "<" in name else []
        P = P_f if g == grid(P_f) else (P_c if g == grid(P_c) else None)
        if base == "mlp_fwd_kernel" and P is not None and targs[2] != "true":
            key = "mlp_fwd_kernel" if targs[3] in ("0", "false") else "mlp_fwd_kernel<save>"
        elif base == "mlp_bwd_chain_kernel" and g == grid(P_f) + grid(P_c):
            key, P = "mlp_bwd_chain_kernel<merged>", P_f + P_c
        elif base == "mlp_bwd_chain_kernel" and P is not None:
            key = "mlp_bwd_chain_kernel"
        elif base in ("mlp_bwd_dw_kernel", "mlp_bwd_dw_f8_kernel"):
            key, P = "mlp_bwd_dw_kernel<merged>", P_f + P_c
        elif base == "mlp_bwd_reduce_kernel":
            key, P = "mlp_bwd_reduce_kernel<merged>", P_f + P_c
        elif base == "mlp_render_kernel" and targs and targs[1] != "0":
            key, P = "mlp_render_kernel<train>", P_f + P_c
        else:
            continue
        db["%s|%s|%d" % (key, a.dtype, P)] = {"hbm_bytes_per_launch": int((2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024),
                                               "FETCH_SIZE_KB": round(cs["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(cs["WRITE_SIZE"], 1)}
    if db:
        note["traffic_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes taken IN THIS RUN on this box (bench.py --pmc-launch: every MLP "
                                "kernel of the step twice; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KB, the guide's gfx950 correction)")
    return db


def load_traffic_db(note):
    """PMC traffic per launch (profiles/pmc_traffic.json, written by tools/pmc_kernels.sh).  The file is stamped with the
    digest of the kernel sources it was measured on; a stamp that does not match the sources of THIS build means stale
    counters: they are then not reported (traffic = null) instead of being passed off as this build's."""
    if not os.path.exists(TRAFFIC_JSON):
        note["traffic_note"] = "no profiles/pmc_traffic.json"
        return {}
    with open(TRAFFIC_JSON) as fh:
        db = json.load(fh)
    from nerf_pl_amd.build import source_digest
    have, want = db.get("_meta", {}).get("source_digest"), source_digest()
    if have != want:
        note["traffic_note"] = "pmc_traffic.json was measured on other kernel sources (digest %s, this build %s): traffic withheld" % (
            str(have)[:12], want[:12])
        return {}
    note["traffic_note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this build's kernel sources (digest %s)" % want[:12]
    return db


def event_time(fn, reps, warm=3, graph=False):
    """Average / min microseconds per call of `fn` by HIP events on torch's current stream (where libnerfhip launches).
    graph=True: `reps` calls are captured into one hipGraph and the replay is timed — for kernels of a few tens of
    microseconds, whose eager issue (ctypes + torch.empty) is slower than the kernel itself."""
    for _ in range(warm):
        fn()
    if graph:
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
        g.replay()
        # same settle as the headline's (main(): the first replays after idle run at ramping clocks): replay for ~60 ms before timing
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        for _ in range(min(200, int(60.0 / max(e0.elapsed_time(e1), 0.05)))):
            g.replay()
        us = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3 / reps)
        return sum(us) / len(us), min(us)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    us = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return sum(us) / len(us), us[0]


def kernel_table(models, rays, S, N, dtype, dev, traffic_db, merged):
    """Per-kernel roofline entries for the MLP kernels of the TIMED training step (fine pass B x (S+N) points and coarse
    pass B x S points): HIP-event time of each kernel on resident buffers, algorithmic FLOPs and HBM bytes
    (DESIGN.md §6), fractions of the dense MFMA peak of the kernel's arithmetic and of the 8 TB/s HBM peak.
    merged: the step runs ONE weight-gradient launch and ONE reduce launch for both models (the fused step at N = 1)."""
    from nerf_pl_amd import _lib, ops
    lib = _lib.load()
    code = ops.mlp_dtype_code(dtype)
    B = rays.shape[0]
    out = []
    with torch.no_grad():
        z = ops.sample_coarse_z(rays, S, False, 0.0)
        zf = ops.fine_z(z, torch.rand(B, S, device=dev), N)

    todo = []                          # (name, tag, P, fn, flops, nbytes, what, key): timed together below, in the step's order

    def entry(name, tag, P, fn, flops, nbytes, what, key_name=None):
        todo.append((name, tag, P, fn, flops, nbytes, what, key_name or name))

    keep, entries, dw_b, P_all, chain_b = [], [], 0, 0, 0
    one_fwd = merged and ops.render_supported(B, S, N, dtype)       # the step's forward is ONE launch (nerfhip_render_train_fwd)
    fwd_bytes = 0
    for tag, model, zz in (("fine pass", models[1], zf), ("coarse pass", models[0], z)):
        P = zz.numel()
        pk = model.packed_weights(dtype)
        pb = model.packed_weights_bwd(dtype)
        acts = ops.alloc_acts(P, dtype, dev)
        raw = ops.mlp_fwd_rays(rays, zz, pk, False, dtype, save=acts)
        g_out = torch.randn_like(raw)
        ws = {}
        ops.mlp_bwd(g_out, raw, pb, acts, dtype, workspace=ws)
        act_b, dy_b = acts.numel(), ws["dys"].numel()
        gate_b = (P + 31) // 32 * 9 * 1024
        # split-K partials the reduce kernel reads: per split 592 used (out-tile, x-tile) blocks of 4 KiB over the 12 jobs
        ws_b = int(lib.nerfhip_mlp_dw_splits(P, code)) // 12 * 592 * 4096
        if not one_fwd:
            entry("mlp_fwd_kernel<save>", tag, P, lambda zz=zz, pk=pk, acts=acts: ops.mlp_fwd_rays(rays, zz, pk, False, dtype, save=acts),
                  FLOP_PER_POINT_FULL * P, act_b + 20 * P, "saved activations + gates written once, 4 B z in + 16 B out per point")
        fwd_bytes += act_b + 56 * P         # + per point: 16 B raw written and read back twice by the compositing waves, 16 B d loss / d raw, z
        if not merged:
            entry("mlp_bwd_chain_kernel", tag, P,
                  lambda g_out=g_out, raw=raw, pb=pb, acts=acts, ws=ws: ops.mlp_bwd(g_out, raw, pb, acts, dtype, phases=1, workspace=ws),
                  FLOP_PER_POINT_DX * P, dy_b + gate_b + 32 * P, "dY written once, ReLU gate words + g_out/out read")
        chain_b += dy_b + gate_b + 32 * P
        if not merged:
            entry("mlp_bwd_dw_kernel", tag, P,
                  lambda g_out=g_out, raw=raw, pb=pb, acts=acts, ws=ws: ops.mlp_bwd(g_out, raw, pb, acts, dtype, phases=2, workspace=ws),
                  FLOP_PER_POINT_DW * P, (act_b - gate_b) + dy_b, "every saved activation and dY slab read once")
            entry("mlp_bwd_reduce_kernel", tag, P,
                  lambda g_out=g_out, raw=raw, pb=pb, acts=acts, ws=ws: ops.mlp_bwd(g_out, raw, pb, acts, dtype, phases=4, workspace=ws),
                  0, ws_b + 2 * 595844 * 4, "split-K partial slabs read, 24 gradient tensors written")
        entries.append((g_out, raw, pb, acts))
        keep.append((acts, raw, g_out, ws))
        dw_b += (act_b - gate_b) + dy_b
        P_all += P
    if one_fwd:
        tgt_ = torch.rand(B, 3, device=dev)
        pr_, u_ = torch.rand(B, S, device=dev), torch.rand(B, N, device=dev)
        gs_ = 2.0 / (3 * B)
        a_c, a_f = keep[1][0], keep[0][0]
        pk_c, pk_f = models[0].packed_weights(dtype), models[1].packed_weights(dtype)
        entry("mlp_render_kernel<train>", "the step's whole forward in ONE launch: coarse + fine MLP, compositing, loss gradient, fine depths, loss",
              P_all, lambda: ops.render_train_fwd(rays, tgt_, gs_, S, N, pk_c, pk_f, dtype, a_c, a_f, False, 1.0, pr_, None, None, 0.0, True, u_),
              FLOP_PER_POINT_FULL * P_all, fwd_bytes,
              "saved activations + gates of both models written once; per point 16 B rgb sigma out and back (L2), 16 B d loss / d raw, depths",
              key_name="mlp_render_kernel<train>")
    if merged:
        wsm = {}
        ops.mlp_bwd_multi(entries, dtype, workspace=wsm)            # (chains included: fills the dY slabs the dW launch reads)
        n_arr = (__import__("ctypes").c_int64 * 2)(*[e[1].numel() // 4 for e in entries])
        n_slabs = int(lib.nerfhip_mlp_dw_workspace_bytes_multi(n_arr, 2, code)) // (4 * (8 * 10 * 64 * 16 + 8 * 64))
        ws_b = n_slabs * 592 * 4096 // 12        # average used blocks per partial slab (592 of a model's 12 jobs together)
        entry("mlp_bwd_chain_kernel", "fine + coarse pass in ONE launch", P_all, lambda: ops.mlp_bwd_multi(entries, dtype, phases=1, workspace=wsm),
              FLOP_PER_POINT_DX * P_all, chain_b, "dY of both models written once, ReLU gate words + g_out/out read", key_name="mlp_bwd_chain_kernel<merged>")
        entry("mlp_bwd_dw_kernel", "fine + coarse pass in ONE launch", P_all, lambda: ops.mlp_bwd_multi(entries, dtype, phases=2, workspace=wsm),
              FLOP_PER_POINT_DW * P_all, dw_b, "every saved activation and dY slab of both models read once", key_name="mlp_bwd_dw_kernel<merged>")
        entry("mlp_bwd_reduce_kernel", "both models in ONE launch", P_all, lambda: ops.mlp_bwd_multi(entries, dtype, phases=4, workspace=wsm),
              0, ws_b + 4 * 595844 * 4, "split-K partial slabs read, 48 gradient tensors written", key_name="mlp_bwd_reduce_kernel<merged>")
    # Timing: each kernel replayed alone (12 launches captured in a hipGraph: no host gaps).  One kernel repeated back to back
    # settles at its own shader clock, which on some boxes is LOWER than inside the step's mix of MFMA-bound and HBM-bound kernels
    # (HIP events between the nodes of one graph do not time on this stack: hipErrorInvalidHandle); the six kernels are therefore
    # also replayed TOGETHER, in the step's order, from one graph: `mix_us` = their time per round in the step's own clock mix.
    order = sorted(range(len(todo)), key=lambda i: (0 if ("fwd" in todo[i][0] and "coarse" in todo[i][1]) or "render" in todo[i][0] else
                                                    1 if "fwd" in todo[i][0] else
                                                    2 if "chain" in todo[i][0] and todo[i][1].startswith("fine") else
                                                    3 if "chain" in todo[i][0] else 4 if "dw" in todo[i][0] else 5, i))
    times = [event_time(todo[i][3], 12, graph=True) for i in order]

    def one_round():
        for i in order:
            todo[i][3]()
    mix_us = event_time(one_round, 4, graph=True)[0]
    # ... and each kernel's duration INSIDE that mix (the clock, power and cache state of the step: what the rocprofv3 trace of the
    # same command shows per kernel).  Events between the nodes of one graph do not time on this stack, so the rounds are issued
    # eagerly with an event between the launches: the host (~20 us per launch) runs far ahead of the GPU (~1 ms per round), the
    # kernels queue back to back and the events stamp their boundaries.
    rounds = 24
    for _ in range(40):                                   # settle (as event_time does) under the same kernel mix
        one_round()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(len(order) + 1)] for _ in range(rounds)]
    for r in range(rounds):
        evs[r][0].record()
        for k, i in enumerate(order):
            todo[i][3]()
            evs[r][k + 1].record()
    torch.cuda.synchronize()
    in_mix = [sum(evs[r][k].elapsed_time(evs[r][k + 1]) for r in range(4, rounds)) * 1e3 / (rounds - 4) for k in range(len(order))]
    for k, i in enumerate(order):
        name, tag, P, _, flops, nbytes, what, key_name = todo[i]
        avg, mn = times[k]
        tf, gbs = flops / avg / 1e6, nbytes / avg / 1e3
        # the dW GEMM of bf16_f8 runs on the MX-scaled fp8 MFMA: priced against ITS dense peak
        peak = PEAK_TFLOPS_FP8 if (dtype == "bf16_f8" and name.startswith("mlp_bwd_dw")) else PEAK_TFLOPS[dtype]
        fm, fh = tf / peak, gbs / PEAK_HBM_GBS
        key = "%s|%s|%d" % (key_name, dtype, P)
        out.append({"kernel": "%s<%s> %s, %d points" % (name, dtype, tag, P), "avg_launch_us": round(avg, 1),
                    "min_launch_us": round(mn, 1), "flops": flops, "hbm_bytes": nbytes, "bytes_are": what,
                    "tflops": round(tf, 1), "gbs": round(gbs, 1), "mfma_peak_tflops": peak, "frac_mfma": round(fm, 4), "frac_hbm": round(fh, 4),
                    # the MLP GEMM kernels are priced against the MFMA roof with the algorithmic FLOPs; `limited_by`
                    # names what this design's kernel actually runs into (its saved-tensor traffic, for the HBM-class ones)
                    "bound": "mfma" if flops else "hbm", "limited_by": "mfma" if fm >= fh else "hbm",
                    "in_step_launch_us": round(in_mix[k], 1),
                    "frac_mfma_in_step": round(flops / max(in_mix[k], 1e-3) / 1e6 / peak, 4),
                    "frac_hbm_in_step": round(nbytes / max(in_mix[k], 1e-3) / 1e3 / PEAK_HBM_GBS, 4),
                    "traffic": traffic_db.get(key, {}).get("hbm_bytes_per_launch")})
    out.sort(key=lambda r: -r["avg_launch_us"])
    del keep, entries, todo
    return out, round(mix_us, 1)


def main():
    a = parse()
    if a.pmc_launch:
        return pmc_launch(a)
    self_launch(a)
    # stdout carries exactly ONE JSON line: libraries that print banners to fd 1 (RCCL prints its version block on the
    # first communicator) are diverted to stderr for the whole run; the result goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    dist, world, rank, local, rccl_nranks = init_world(a)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from nerf_pl_amd import ops
    from nerf_pl_amd.models import rendering
    from nerf_pl_amd.parallel import GradSync
    from nerf_pl_amd.system import GraphedTrainStep, NeRFSystem

    B, S, N = a.rays, a.n_samples, a.n_importance
    hp = Namespace(N_samples=S, N_importance=N, use_disp=False, perturb=1.0, noise_std=0.0, chunk=1024 * 32,
                   loss_type="mse", lr=5e-4, weight_decay=0, decay_step=[2, 4, 8], decay_gamma=0.5, white_back=True,
                   optimizer="adam", lr_scheduler="steplr")

    def build_system(dtype, hp=hp):
        system = NeRFSystem(hp)
        # random-init weights of the named architecture (identical on every rank = DDP replicas), density head
        # scaled so that opacity is non-trivial
        system.nerf_coarse.load_state_dict(synth_params(100, 4.0, 0.2))
        system.nerf_fine.load_state_dict(synth_params(101, 4.0, 0.2))
        for m in system.models:
            m.mlp_dtype = dtype
        system = system.to(dev)
        (opt,), _ = system.configure_optimizers()
        system.fused_train_step = not a.modular_step
        # one rank: no all-reduce sits between the gradients and the update, so the reduce kernel applies Adam in place
        system.fuse_adam = (dist is None) and a.fuse_adam and not a.no_fuse_adam and not a.modular_step
        return system, opt

    system, opt = build_system(a.dtype)
    models, emb = system.models, system.embeddings
    grad_sync = GradSync(models, force=a.force_dist) if dist is not None else None
    rays = synth_rays(1234 + rank, B).to(dev)                 # fixed batch: render mode, per-kernel timings
    store = synth_store(4321 + rank, dev)                     # each rank owns its own images and draws its own batches
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    torch.manual_seed(rank)

    def next_batch():
        if a.fixed_batch:
            return {"rays": rays, "rgbs": fixed_rgbs}
        return store.sample(B, generator=gen)                 # randint + gen_rays + gather on the GPU, inside the timed loop
    fixed_rgbs = torch.rand(B, 3, generator=torch.Generator().manual_seed(rank)).to(dev)

    def render_step():
        with torch.no_grad():
            return rendering.render_rays(models, emb, rays, S, False, 1.0, 0.0, N, 1024 * 32, True)

    # The training step (fwd, loss, bwd, [all-reduce], Adam: ~40 launches) is replayed as ONE hipGraph after 3 eager
    # steps; same work per step, ~15 us of host time instead of ~1.5 ms.  Falls back to eager issue if capture fails.
    def make_stepper(system_, opt_, sync_, store_=None):
        # fresh batches are drawn INSIDE the graph: RayStore.sample draws the pixel ids from the default generator's Philox stream,
        # generates their rays, gathers their colours AND makes the step's four render_rays draws in ONE captured launch; the
        # generator state lives on the device and advances per replay (nerf_pl_amd/draws.py): no per-step copies into static buffers
        in_graph = not a.fixed_batch
        store_ = store_ if store_ is not None else store
        hp_ = system_.hp
        kw = {} if a.sync_in_graph is None else {"sync_in_graph": bool(a.sync_in_graph)}
        from nerf_pl_amd.system import _HipGraphBackend
        # ... and packs both models' weight images in that same launch (nerfhip_train_prologue): the step's whole prologue is one node
        pk = (system_.models, system_.models[0].mlp_dtype) if system_.fused_train_step else None
        src = (lambda: store_.sample(B, step_draws=(hp_.N_samples, hp_.N_importance, hp_.perturb, hp_.noise_std), pack_models=pk)) if in_graph else None
        st = {"graphed": GraphedTrainStep(system_, opt_, sync_, warmup=3, batch_source=src, backend=_HipGraphBackend(keep_graph=True), **kw)
              if not a.no_graph else None}

        def eager(batch):
            out = system_.training_step(batch, 0)
            opt_.zero_grad(set_to_none=True)
            out["loss"].backward()
            if sync_ is not None:
                sync_.sync()
            opt_.step()
            return out

        def step_():
            g = st["graphed"]
            if g is None:
                return eager(next_batch())
            try:
                return g() if g.batch_source is not None else g(next_batch())
            except Exception as e:  # noqa: BLE001 - capture problems must not kill the benchmark
                print("[bench] hipGraph capture failed (%s: %s); continuing eagerly" % (type(e).__name__, e), file=sys.stderr, flush=True)
                st["graphed"] = None
                torch.cuda.synchronize()
                return eager(next_batch())
        return step_, st

    train_step, state = make_stepper(system, opt, grad_sync)

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
        # one rank: finished chunks stream to pinned host memory while the next chunk replays (the image ends up where
        # eval.py:123 needs it: on the host); N ranks: ray-sharded render, pixels all-gathered, rank 0 copies them out
        if world == 1:
            return eval_state["gr"].render_to_host(eval_state["rays"], keys=("rgb_fine", "depth_fine"))
        res = render_sharded(eval_state["gr"], eval_state["rays"], keys=("rgb_fine", "depth_fine"))
        return {k: v.cpu() for k, v in res.items()} if rank == 0 else res

    step = train_step if a.mode == "train" else (eval_step if a.mode == "eval" else render_step)

    series = [] if a.series else None

    def timed(step_fn, warmup, steps):
        return timed_region(step_fn, warmup, steps, dist, dev, series=series)

    def side_steps():
        """Extras of the default line, measured AFTER the headline's timed region: the fp8-dW variant of the same step (its own
        process: a dedicated run, not a second system squeezed into this one), BASELINE configs[1] (fp32, 64+64) and configs[3]
        (NDC rays, noise_std=1, black background, 64+64) training steps."""
        import subprocess
        ex = {}
        step_flops_pt = FLOP_PER_POINT_FULL + FLOP_PER_POINT_DX + FLOP_PER_POINT_DW
        if a.dtype == "bf16":
            cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "bf16_f8", "--steps", "15", "--warmup", "6", "--no-extras",
                   "--no-cpu-baseline", "--rays", str(B), "--n-samples", str(S), "--n-importance", str(N)]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=150)
                line = json.loads(r.stdout.strip().splitlines()[-1])
                ex["f8_dw_ms_per_step"] = line["ms_per_step"]
                ex["f8_dw_rays_per_s"] = line["value"]
                ex["f8_dw_dtype"] = line["dtype"]
                ex["f8_dw_step_frac_mfma_of_bf16_peak"] = line.get("step_frac_mfma")
                ex["f8_dw_mlp_kernels_us_per_step"] = line.get("mlp_kernels_us_per_step")
                ex["roofline_kernels_f8"] = line.get("roofline_kernels")
                ex["f8_dw_note"] = ("the same step with the saved activations / dY stored as block-scaled 8-bit floats and the dW GEMM on "
                                    "the MX-scaled fp8 MFMA (licensed by tests/test_gpu_psnr_gate.py), measured by its own "
                                    "`bench.py --dtype bf16_f8` process after the headline; NOT the BASELINE-named arithmetic")
            except Exception as e:  # noqa: BLE001 - an extra, never fatal
                ex["f8_dw_note"] = "fp8-dW side run failed: %s: %s" % (type(e).__name__, e)
        # configs[1]: fp32 (the 1e-4 parity arithmetic), 64 + 64 samples
        hp1 = Namespace(**dict(vars(hp), N_importance=64))
        sys1, opt1 = build_system("fp32", hp1)
        st1, _ = make_stepper(sys1, opt1, None)
        t1 = timed(st1, 5, 8) / 8
        ex["fp32_c1_ms_per_step"] = round(t1 * 1e3, 4)
        ex["fp32_c1_frac_mfma"] = round(step_flops_pt * B * (2 * S + 64) / t1 / 1e12 / PEAK_TFLOPS["fp32"], 4)
        ex["fp32_c1_note"] = "configs[1]: %d rays x (%d+64) samples, exact-fp32 MFMA MLP, full training step; frac of the 157.3 TFLOP/s fp32 MFMA peak" % (B, S)
        del sys1, opt1, st1
        # configs[3]: LLFF-style NDC rays (non-unit directions), noise_std = 1 (rendering.py:152 noise path), black background, 64 + 64
        hp3 = Namespace(**dict(vars(hp), N_importance=64, noise_std=1.0, white_back=False))
        sys3, opt3 = build_system(a.dtype, hp3)
        st3, _ = make_stepper(sys3, opt3, None, synth_store_ndc(777, dev))
        t3 = timed(st3, 5, 15) / 15
        ex["ndc_c3_ms_per_step"] = round(t3 * 1e3, 4)
        ex["ndc_c3_frac_mfma"] = round(step_flops_pt * B * (2 * S + 64) / t3 / 1e12 / PEAK_TFLOPS[a.dtype], 4)
        ex["ndc_c3_note"] = ("configs[3] per GPU: %d NDC rays x (%d+64) samples, noise_std=1, white_back=False, %s, full training step "
                             "(the 8-GPU half of configs[3] is the --gpus N line)" % (B, S, DTYPE_LABEL[a.dtype]))
        del sys3, opt3, st3
        return ex

    # Setup, before the contract's W untimed + K timed steps:
    #  (i)  the step is BUILT: the stepper's 3 eager steps (autograd state, allocations) and the capture of its hipGraph.  Until
    #       round 4 these four calls were counted as the first four of the W warm-up steps (W = 5 left one replay);
    #  (ii) the device is brought to its SUSTAINED state: right after the capture the first ~60 replays run 5-15 % slower than the
    #       ones after them (profiles/r05_replay_series.txt: 1.28-1.33 ms for replays 1-5, 1.20 for 6-15, 1.13-1.14 from ~70 on;
    #       the host-bound eager steps leave the GPU mostly idle and its clocks low).  A training run is 10^5 such steps, so the
    #       metric is the sustained rate; the first K replays after the capture are timed too and reported beside it
    #       (`cold_start_ms_per_step`), then --settle replays run untimed (default: by mode; 0 = none).
    # Every rank runs the same counts (the collectives of an N > 1 step stay matched).
    cold_ms = None
    settle = a.settle if a.settle is not None else {"train": 150, "render": 300, "eval": 1}[a.mode]
    if a.mode == "train" and not a.no_graph:
        for _ in range(4):                                   # 3 eager steps + capture (and its first replay)
            step()
    if settle > 0:
        k_cold = min(a.steps, settle)
        cold_ms = timed_region(step, 0, k_cold, dist, dev) / k_cold * 1e3
        for _ in range(settle - k_cold):
            step()
    dt = timed(step, a.warmup, a.steps)
    if series is not None and rank == 0:
        print("[bench] per-step ms of the timed steps: %s" % json.dumps(series), file=sys.stderr, flush=True)
        series = None

    if rank == 0:
        extra = {}
        traffic_db = load_traffic_db(extra)
        if a.mode == "train" and world == 1 and not a.no_extras and not a.no_pmc and a.dtype != "fp32":
            live = pmc_collect(a, extra)                     # this box's own counters replace the stamped file's
            if live:
                traffic_db = dict(traffic_db, **live)
        if a.mode == "train":                                  # forward-only rate of the same workload
            for _ in range(5):
                render_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(30):
                render_step()
            torch.cuda.synchronize()
            t_render = (time.perf_counter() - t1) / 30
            extra["render_fwd_rays_per_s_per_gpu"] = round(B / t_render, 1)
            # train-mode forward = full coarse + full fine network: FLOP_PER_POINT_FULL x (2S + N) points per ray
            extra["render_fwd_frac_mfma"] = round(FLOP_PER_POINT_FULL * B * (2 * S + N) / t_render / 1e12 / PEAK_TFLOPS[a.dtype], 4)
            if world == 1 and not a.no_extras:
                # BASELINE configs[4] inside the default run: full 800x800 images through eval.py's batched_inference contract
                # (test_time, 32768-ray hipGraph chunks, pixels streamed to pinned host memory: D2H inside the timed region)
                from nerf_pl_amd.inference import GraphRenderer
                e_rays = synth_rays(77, a.image_rays).to(dev)
                gr = GraphRenderer(models, emb, S, N, False, True)
                gr.render_to_host(e_rays, keys=("rgb_fine", "depth_fine"))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    gr.render_to_host(e_rays, keys=("rgb_fine", "depth_fine"))
                torch.cuda.synchronize()
                t_img = (time.perf_counter() - t1) / 3
                extra["eval_ms_per_image"] = round(t_img * 1e3, 2)
                extra["eval_rays_per_s"] = round(a.image_rays / t_img, 1)
                extra["eval_frac_mfma"] = round(FLOP_PER_RAY_EVAL * a.image_rays / t_img / 1e12 / PEAK_TFLOPS[a.dtype], 4)
                extra["eval_note"] = ("configs[4]: %d-ray image, 32768-ray hipGraph chunks x (%d+%d) samples, test_time (sigma-only coarse "
                                      "pass), D2H of rgb+depth included; 3 images after 1 warm-up" % (a.image_rays, S, N))
                del gr, e_rays

        # ---- the north-star kernel: fine-pass fused MLP forward, inference variant (B x (S+N) points) ----
        with torch.no_grad():
            z = ops.sample_coarse_z(rays, S, False, 0.0)
            zf = ops.fine_z(z, torch.rand(B, S, device=dev), N)
            pk = models[1].packed_weights()                     # pack once: the events bracket the MLP kernel alone
            avg_us, min_us = event_time(lambda: ops.mlp_fwd_rays(rays, zf, pk, False, a.dtype), 30, warm=5)
        flops = FLOP_PER_POINT_FULL * B * (S + N)
        ach = flops / avg_us / 1e6
        ns = {"bound": "mfma", "kernel": "mlp_fwd_kernel<%s,rays,full> (inference) %dx%d points" % (a.dtype, B, S + N),
              "achieved": round(ach, 2), "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
              "frac": round(ach / PEAK_TFLOPS[a.dtype], 4),
              "traffic": traffic_db.get("mlp_fwd_kernel|%s|%d" % (a.dtype, B * (S + N)), {}).get("hbm_bytes_per_launch"),
              "avg_launch_us": round(avg_us, 2), "min_launch_us": round(min_us, 2)}
        if a.dtype != "fp32":
            # informational: what this chip sustains at all under back-to-back bf16 MFMAs (it is power-limited: the shader
            # clock settles at 1.45-1.83 GHz), measured by tools/probes/probe_mfma_rate.hip -> profiles/archive/r02_probe_mfma_rate.txt
            ns["measured_ceiling"] = {"bare_mfma_frac_of_peak": [0.61, 0.66], "this_instruction_mix_frac_of_peak": [0.57, 0.59],
                                      "source": "profiles/archive/r02_probe_mfma_rate.txt"}
        if a.mode == "train":
            # ---- the kernels the TIMED step runs, one entry each; `roofline` = the one that takes the most time ----
            table, mix_us = kernel_table(models, rays, S, N, a.dtype, dev, traffic_db, merged=(system.fused_train_step and grad_sync is None))
            dom = max(table, key=lambda r: r["in_step_launch_us"])
            # The headline figure follows SURVEY 8(d): an MLP kernel is priced against the dense MFMA peak of its arithmetic with the
            # ALGORITHMIC FLOPs of the GEMMs it performs.  The HBM view of the same launch (this design materialises activations
            # and dY for the weight-gradient GEMM: bytes that 8(d)'s fused arithmetic intensity does not contain) rides beside it.
            # Durations: the kernel's time INSIDE the step's kernel mix (`in_step_launch_us`: what the rocprofv3 trace of this command
            # shows for it); the same launch repeated alone settles at its own clock and runs a few % faster (`alone_launch_us`).
            us = dom["in_step_launch_us"]
            gbs_, tf_ = dom["hbm_bytes"] / us / 1e3, dom["flops"] / us / 1e6
            roof = {"bound": dom["bound"], "kernel": dom["kernel"],
                    "achieved": round(gbs_ if dom["bound"] == "hbm" else tf_, 1),
                    "peak": PEAK_HBM_GBS if dom["bound"] == "hbm" else dom["mfma_peak_tflops"],
                    "unit": "GB/s" if dom["bound"] == "hbm" else "TFLOP/s",
                    "frac": dom["frac_hbm_in_step"] if dom["bound"] == "hbm" else dom["frac_mfma_in_step"],
                    "traffic": dom["traffic"], "avg_launch_us": us, "alone_launch_us": dom["avg_launch_us"],
                    "frac_mfma": dom["frac_mfma_in_step"], "frac_hbm": dom["frac_hbm_in_step"], "limited_by": dom["limited_by"],
                    "hbm_view": {"achieved": round(gbs_, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": dom["frac_hbm_in_step"],
                                 "bytes": dom["hbm_bytes"], "bytes_are": dom["bytes_are"]}}
            extra["roofline"] = roof
            extra["roofline_kernels"] = table
            extra["roofline_north_star"] = ns
            # the step's MLP kernels replayed together in step order from one graph (<= ms_per_step * 1000) / summed from the
            # kernel-alone entries (each at its own clock: may come out above or below the former, box dependent)
            extra["mlp_kernels_us_per_step"] = mix_us
            extra["mlp_kernels_us_alone_sum"] = round(sum(r["avg_launch_us"] for r in table), 1)
            # whole-step MFMA fraction: algorithmic FLOPs of the step (GEMMs only) over the step time
            step_flops = (FLOP_PER_POINT_FULL + FLOP_PER_POINT_DX + FLOP_PER_POINT_DW) * B * (2 * S + N)
            extra["step_frac_mfma"] = round(step_flops / (dt / a.steps) / 1e12 / PEAK_TFLOPS[a.dtype], 4)
            g_ = state["graphed"]
            if g_ is not None and g_.graph is not None:
                from nerf_pl_amd.system import graph_node_count
                # nodes of the replayed hipGraph = launches per step (hipGraphGetNodes on the captured graph)
                extra["launches_per_step"] = graph_node_count(g_.graph)
            extra["non_mlp_us"] = round(dt / a.steps * 1e6 - mix_us, 1)      # step - the six MLP kernels replayed together
            if world == 1 and not a.no_extras:
                extra.update(side_steps())
        else:
            extra["roofline"] = ns

        total_rays = (a.image_rays if a.mode == "eval" else world * B) * a.steps
        arith = DTYPE_LABEL[a.dtype]
        mlp_note = {"bf16_f8": "bf16 MFMA MLP (forward + dX chain), dW GEMM on block-scaled e4m3 copies of the saved tensors",
                    "bf16": "bf16 MFMA MLP", "fp32": "exact-fp32 MFMA MLP"}[a.dtype]
        out = {
            "metric": ("rays/sec (64+128 samples), full training step: render_rays fwd + MSE + bwd + grad all-reduce + Adam"
                       if a.mode == "train" else
                       "rays/sec (64+128 samples), full-image inference incl. D2H of the pixels (eval.py batched_inference, test_time, 32768-ray hipGraph chunks)"
                       if a.mode == "eval" else
                       "rays/sec (64+128 samples), render_rays forward only (train-mode: coarse+fine rgb)"),
            "value": round(total_rays / dt, 1), "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if a.mode == "eval" else "weak",
            "vs_baseline": None, "dtype": arith, "data": "synthetic",
            "config": {"workload": ("configs[4]: %d-ray image in 32768-ray chunks x (%d+%d) samples, test_time, NeRF D8 W256 "
                                    "coarse(sigma-only)+fine, %s, mode=eval" % (a.image_rays, S, N, mlp_note))
                                   if a.mode == "eval" else
                                   ("configs[2]: %d rays/GPU x (%d+%d) samples, NeRF D8 W256 coarse+fine, perturb=1 "
                                    "noise_std=0 white_back, %s, mode=%s" % (B, S, N, mlp_note, a.mode)),
                       "mlp_dtype": a.dtype,
                       "rays_per_gpu": (a.image_rays // world if a.mode == "eval" else B), "N_samples": S, "N_importance": N,
                       "batches": ("one resident batch replayed" if (a.fixed_batch or a.mode != "train") else
                                   "fresh RayStore.sample(%d) per step inside the timed loop (pixel ids + rays + the step's draws in one launch on the GPU%s)"
                                   % (B, ", captured in the step's hipGraph" if not a.no_graph else "")),
                       "issue": ("hipGraph replay of the whole step" if (a.mode == "train" and state["graphed"] is not None
                                                                         and state["graphed"].graph is not None)
                                 else "hipGraph replay per 32768-ray chunk" if a.mode == "eval" else "eager"),
                       "parallelism": "ray-sharded x%d%s" % (world, ", RCCL grad all-reduce" if dist is not None and a.mode == "train" else ""),
                       "step_form": (None if a.mode != "train" else
                                     "modular autograd graph (render_rays -> MSELoss), separate Adam launch" if a.modular_step else
                                     ("fused node: batch + draws + both weight packs in one launch; the whole forward in ONE launch (workgroups own 4 rays: "
                                      "coarse MLP, compositing + loss gradient + compositing backward + fine depths, fine MLP, the same + loss values); "
                                      "one chain / dW / reduce launch for both models" if ops.render_supported(B, S, N, a.dtype) else
                                      "fused node: batch + draws + both weight packs in one launch, coarse depths in the MLP prologue, composite+loss-gradient+"
                                      "composite-backward (+ fine depths | + loss) per pass, one chain / dW / reduce launch for both models")
                                     + (", Adam applied inside the reduce kernel" if system.fuse_adam else ", separate Adam launch")),
                       "rccl_nranks": rccl_nranks,
                       "capture_fallback": (getattr(state["graphed"], "capture_fallback", None) if a.mode == "train" else None),
                       "grad_sync": (None if (grad_sync is None or a.mode != "train") else
                                     ("one hipGraph, all-reduces issued from the grad-ready hooks inside it"
                                      if (state["graphed"] is not None and not state["graphed"]._two_graphs()) else
                                      "two hipGraphs (fwd+bwd | Adam), flat-buffer all-reduces issued eagerly in between"
                                      if state["graphed"] is not None else "eager, hook-overlapped all-reduces"))},
        }
        out["setup"] = {"build_calls_before_warmup": 4 if (a.mode == "train" and not a.no_graph) else 0,
                        "settle_replays_before_warmup": settle,
                        "note": "outside W + K: 3 eager steps + hipGraph capture, then `settle_replays` untimed replays (the first ~60 "
                                "replays after a cold start run 5-15 % slower: device clocks); the first K of them are timed as "
                                "cold_start_ms_per_step; --settle 0 measures right after the capture"}
        if cold_ms is not None:
            out["cold_start_ms_per_step"] = round(cold_ms, 4)
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
