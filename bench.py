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
from bench_inputs import (DTYPE_LABEL, PARITY, FLOP_PER_POINT_DW, FLOP_PER_POINT_DX, FLOP_PER_POINT_FULL, FLOP_PER_RAY_EVAL, FLOP_PER_POINT_FULL_EXECUTED,
                          FLOP_PER_POINT_DX_EXECUTED, FLOP_PER_POINT_DW_EXECUTED,  # noqa: E402
                          PEAK_HBM_GBS, PEAK_TFLOPS, cpu_baseline, synth_params, synth_rays, synth_store, synth_store_ndc)

PROTOCOL_VERSION = 3      # 1: rounds 1-3 (build calls counted as warm-up); 2: rounds 4-5 (build + settle replays outside W + K, value = sustained,
                          # cold_start_ms_per_step = the first K replays); 3: round 6 (`literal_contract` = exactly W untimed + K timed right
                          # after the build, then the settle replays, then W + K again = `value`; both on every line)


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
    ap.add_argument("--workload", default="c2", choices=["c2", "c3"],
                    help="c2 (default): BASELINE configs[2], Blender-style rays, 64+128 samples, noise_std 0, white background — the configuration "
                         "the metric is quoted on.  c3: configs[3] per GPU — LLFF-style NDC rays (non-unit directions), 64+64 samples, noise_std 1, "
                         "black background — as the MAIN step (so that tools/ktrace_step.sh / the PMC passes can look at it); a parity / "
                         "diagnosis line, not the headline")
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--n-samples", type=int, default=64)
    ap.add_argument("--n-importance", type=int, default=None, help="default 128 (workload c2) / 64 (c3)")
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
                    help="N>1 graphed step: 1 = ONE graph with the hook-issued all-reduce(s) inside (the default since round 6; falls back to "
                         "two graphs, agreed across the ranks, when a collective cannot be captured), 0 = two graphs with the collectives "
                         "issued eagerly in between")
    ap.add_argument("--grad-sync-form", default=None, choices=["merged", "per_model"],
                    help="N>1: merged (default) = the one-rank launches + ONE all-reduce of the step's joint gradient buffer; per_model = "
                         "the two models' backwards one after the other, the fine model's all-reduce under the coarse model's backward")
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

def main():
    a = parse()
    if a.n_importance is None:
        a.n_importance = 64 if a.workload == "c3" else 128
    if a.pmc_launch:
        from bench_extras import pmc_launch
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
    c3 = a.workload == "c3"
    hp = Namespace(N_samples=S, N_importance=N, use_disp=False, perturb=1.0, noise_std=1.0 if c3 else 0.0, chunk=1024 * 32,
                   loss_type="mse", lr=5e-4, weight_decay=0, decay_step=[2, 4, 8], decay_gamma=0.5, white_back=not c3,
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
    grad_sync = GradSync(models, force=a.force_dist, form=a.grad_sync_form) if dist is not None else None
    rays = synth_rays(1234 + rank, B).to(dev)                 # fixed batch: render mode, per-kernel timings
    # each rank owns its own images and draws its own batches
    store = synth_store_ndc(777 + rank, dev) if c3 else synth_store(4321 + rank, dev)
    if c3:                                                    # the fixed batch of render mode / the per-kernel timings: NDC rays of that store
        rays = store.sample(B, generator=torch.Generator(device=dev).manual_seed(5))["rays"].contiguous()
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
            return rendering.render_rays(models, emb, rays, S, False, 1.0, hp.noise_std, N, 1024 * 32, hp.white_back)

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

    # The measurement protocol (PROTOCOL_VERSION 3).  Every rank runs the same counts (the collectives of an N > 1 step stay matched).
    #  (i)   BUILD, outside everything: the stepper's 3 eager steps (autograd state, allocations) + the capture of its hipGraph.
    #  (ii)  `literal_contract`: EXACTLY the contract's W untimed + K timed steps, right after the build — the device as the build left
    #        it (the host-bound eager steps leave the GPU mostly idle and its clocks low: the first ~60 replays run 5-15 % slower than
    #        the ones after them, profiles/r05_replay_series.txt).
    #  (iii) settle: untimed replays until `--settle` replays have run since the build (default by mode; 0 = none: (ii) IS the value).
    #  (iv)  `value`: W untimed + K timed steps again, on the device in its SUSTAINED state — a training run is 10^5 such steps.
    # Both figures are on every line; rounds 4-5 printed (iv) as `value` and the first K replays WITHOUT warm-up as
    # `cold_start_ms_per_step` (protocol 2); profiles/README.md has the r4 / r5 / r6 builds under one harness.
    settle = a.settle if a.settle is not None else {"train": 150, "render": 300, "eval": 1}[a.mode]
    if a.mode == "train" and not a.no_graph:
        for _ in range(4):                                   # 3 eager steps + capture (and its first replay)
            step()
    dt_literal = timed_region(step, a.warmup, a.steps, dist, dev)
    if settle > 0:
        for _ in range(max(0, settle - a.warmup - a.steps)):
            step()
        dt = timed(step, a.warmup, a.steps)
    else:
        dt = dt_literal
    if series is not None and rank == 0:
        print("[bench] per-step ms of the timed steps: %s" % json.dumps(series), file=sys.stderr, flush=True)
        series = None

    if rank == 0:
        extra = {}
        from bench_extras import event_time, kernel_table, load_traffic_db
        traffic_db = load_traffic_db(extra)
        if a.mode == "train" and world == 1 and not a.no_extras and not a.no_pmc and a.dtype != "fp32":
            from bench_extras import pmc_collect
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
              "avg_launch_us": round(avg_us, 2), "min_launch_us": round(min_us, 2),
              # `frac` prices the reference's GEMMs (1,186,816 FLOP / point); the kernel executes 1,055,744 of them per point:
              # xyz_encoding_final has no activation and is folded into the dir layer (csrc/mlp_layout.h kLayers)
              "frac_executed": round(FLOP_PER_POINT_FULL_EXECUTED * B * (S + N) / avg_us / 1e6 / PEAK_TFLOPS[a.dtype], 4)}
        if a.dtype != "fp32":
            # informational: what this chip sustains at all under back-to-back bf16 MFMAs (it is power-limited: the shader
            # clock settles at 1.45-1.83 GHz), measured by tools/probes/probe_mfma_rate.hip -> profiles/archive/r02_probe_mfma_rate.txt
            ns["measured_ceiling"] = {"bare_mfma_frac_of_peak": [0.61, 0.66], "this_instruction_mix_frac_of_peak": [0.57, 0.59],
                                      "source": "profiles/archive/r02_probe_mfma_rate.txt"}
        if a.mode == "train":
            # ---- the kernels the TIMED step runs, one entry each; `roofline` = the one that takes the most time ----
            table, mix_us = kernel_table(models, rays, S, N, a.dtype, dev, traffic_db, merged=(system.fused_train_step and (grad_sync is None or grad_sync.form == "merged")))
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
            # ... and with the FLOPs the kernels execute (the final layer folded in forward, chain and dW: 88.7 % of the algorithmic figure)
            extra["step_frac_mfma_executed"] = round((FLOP_PER_POINT_FULL_EXECUTED + FLOP_PER_POINT_DX_EXECUTED + FLOP_PER_POINT_DW_EXECUTED)
                                                     * B * (2 * S + N) / (dt / a.steps) / 1e12 / PEAK_TFLOPS[a.dtype], 4)
            g_ = state["graphed"]
            if g_ is not None and g_.graph is not None:
                from nerf_pl_amd.system import graph_node_count
                # nodes of the replayed hipGraph = launches per step (hipGraphGetNodes on the captured graph)
                extra["launches_per_step"] = graph_node_count(g_.graph)
            extra["non_mlp_us"] = round(dt / a.steps * 1e6 - mix_us, 1)      # step - the six MLP kernels replayed together
            if world == 1 and not a.no_extras:
                from bench_extras import side_steps
                extra.update(side_steps(a, hp, build_system, make_stepper, timed, dev))
        else:
            extra["roofline"] = ns

        total_rays = (a.image_rays if a.mode == "eval" else world * B) * a.steps
        arith = DTYPE_LABEL[a.dtype]
        mlp_note = {"bf16_f8": "bf16 MFMA MLP (forward + dX chain), dW GEMM on block-scaled e4m3 copies of the saved tensors",
                    "bf16": "bf16 MFMA MLP", "fp32": "exact-fp32 MFMA MLP"}[a.dtype]
        out = {
            "metric": ("rays/sec (%d+%d samples), full training step: render_rays fwd + MSE + bwd + grad all-reduce + Adam"
                       if a.mode == "train" else
                       "rays/sec (%d+%d samples), full-image inference incl. D2H of the pixels (eval.py batched_inference, test_time, 32768-ray hipGraph chunks)"
                       if a.mode == "eval" else
                       "rays/sec (%d+%d samples), render_rays forward only (train-mode: coarse+fine rgb)") % (S, N),
            "value": round(total_rays / dt, 1), "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if a.mode == "eval" else "weak",
            "vs_baseline": None, "dtype": arith, "data": "synthetic",
            "config": {"workload": ("configs[4]: %d-ray image in 32768-ray chunks x (%d+%d) samples, test_time, NeRF D8 W256 "
                                    "coarse(sigma-only)+fine, %s, mode=eval" % (a.image_rays, S, N, mlp_note))
                                   if a.mode == "eval" else
                                   ("configs[3] per GPU: %d LLFF-style NDC rays/GPU x (%d+%d) samples, NeRF D8 W256 coarse+fine, perturb=1 "
                                    "noise_std=1 black background, %s, mode=%s" % (B, S, N, mlp_note, a.mode)) if c3 else
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
                                     (("one hipGraph, all-reduce(s) issued from the backward's hook inside it"
                                       if (state["graphed"] is not None and not state["graphed"]._two_graphs()) else
                                       "two hipGraphs (fwd+bwd | Adam), flat-buffer all-reduce(s) issued eagerly in between"
                                       if state["graphed"] is not None else "eager, hook-issued all-reduce(s)")
                                      + ("; form=merged: the one-rank launches + ONE all-reduce over the step's joint gradient buffer (4.77 MB)"
                                         if grad_sync.form == "merged" else
                                         "; form=per_model: per model chain -> dW -> reduce -> all-reduce (2.38 MB each)")))},
        }
        per_step_rays = a.image_rays if a.mode == "eval" else world * B
        lit = {"ms_per_step": round(dt_literal / a.steps * 1e3, 4), "value": round(per_step_rays * a.steps / dt_literal, 1), "unit": "rays/s",
               "warmup": a.warmup, "steps": a.steps,
               "what": "exactly W untimed + K timed steps right after the build (3 eager steps + hipGraph capture), nothing in between"}
        if a.mode == "train":
            lit["step_frac_mfma"] = round((FLOP_PER_POINT_FULL + FLOP_PER_POINT_DX + FLOP_PER_POINT_DW) * B * (2 * S + N)
                                          / (dt_literal / a.steps) / 1e12 / PEAK_TFLOPS[a.dtype], 4)
        out["literal_contract"] = lit
        out["protocol"] = {"version": PROTOCOL_VERSION,
                           "value_is": "sustained: W + K after %d settle replays" % settle if settle > 0 else "literal_contract (--settle 0)",
                           "build_calls_before_warmup": 4 if (a.mode == "train" and not a.no_graph) else 0,
                           "settle_replays_since_build": settle,
                           "note": "build (3 eager steps + capture) -> literal_contract (W + K) -> untimed replays up to `settle` since the build -> "
                                   "value (W + K).  Rounds 4-5 (protocol 2) printed the same sustained `value` and the first K replays without "
                                   "warm-up as cold_start_ms_per_step; rounds 1-3 (protocol 1) counted the build calls as warm-up steps"}
        out.update(extra)
        out["parity"] = PARITY
        if not a.no_cpu_baseline and world == 1:                # reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(B, S, N, a.cpu_seconds, a.mode == "train")
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
