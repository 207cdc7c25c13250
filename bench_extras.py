"""bench_extras.py — the side measurements of bench.py's default line (split out of bench.py in round 6, no behaviour change).

Nothing here is inside the contract's timed region (that is bench.py: `timed_region`).  What lives here:
  * `pmc_launch` / `pmc_collect` / `load_traffic_db` — HBM counters of THIS run's kernels (rocprofv3 --pmc, separate FETCH_SIZE /
    WRITE_SIZE passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled: gfx950 tallies 64 B per
    128-B request of the wide reads these kernels make).  The passes re-run bench.py in `--pmc-launch` mode.
  * `event_time` / `kernel_table` — HIP-event durations of the step's MLP kernels (alone, and inside the step's kernel mix) with their
    algorithmic FLOPs / bytes: the `roofline` / `roofline_kernels` objects of the line.
"""
import json
import os
import sys

import torch

from bench_inputs import (FLOP_PER_POINT_DW, FLOP_PER_POINT_DW_EXECUTED, FLOP_PER_POINT_DX_EXECUTED, FLOP_PER_POINT_FULL_EXECUTED, UNSAVED_SLABS_PER_TILE,
                          FLOP_PER_POINT_DX, FLOP_PER_POINT_FULL, PEAK_HBM_GBS, PEAK_TFLOPS, PEAK_TFLOPS_FP8, ROOT,
                          TRAFFIC_JSON, synth_params, synth_rays)

BENCH_PY = os.path.join(ROOT, "bench.py")


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
    from nerf_pl_amd.models.train_step import regen_enc_enabled
    # (as the step does: bf16 leaves the input encodings out of the saved activations, the dW launch forms them again)
    regen = regen_enc_enabled() and a.dtype == "bf16" and ops.render_supported(B, S, N, a.dtype) and S % 32 == 0 and (S + N) % 32 == 0
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
            entries.append((torch.randn_like(raw), raw, pb, acts) + ((rays, zz) if regen else ()))
        ws = {}
        for _ in range(2):
            ops.mlp_bwd_multi(entries, a.dtype, workspace=ws)           # chain x 2, merged dW, merged reduce
            ops.mlp_fwd_rays(rays, zf, pk, False, a.dtype)               # the inference forward (north star)
        if ops.render_supported(B, S, N, a.dtype):                       # the step's forward as ONE launch
            tgt, pr, u = torch.rand(B, 3, device=dev), torch.rand(B, S, device=dev), torch.rand(B, N, device=dev)
            pk_c = models[0].packed_weights(a.dtype)
            for _ in range(2):
                ops.render_train_fwd(rays, tgt, 2.0 / (3 * B), S, N, pk_c, pk, a.dtype, entries[1][3], entries[0][3], False, 1.0, pr, None, None,
                                     0.0, True, u, regen_enc=regen)
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
            cmd = ["rocprofv3", "--pmc", C, "--kernel-trace", "-f", "csv", "-d", out, "-o", "p", "--", sys.executable, BENCH_PY,
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
    (DESIGN.md §7), fractions of the dense MFMA peak of the kernel's arithmetic and of the 8 TB/s HBM peak.
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

    executed = {}                      # key -> FLOPs the launch executes, where that is less than the algorithmic figure

    def entry(name, tag, P, fn, flops, nbytes, what, key_name=None, flops_executed=None):
        todo.append((name, tag, P, fn, flops, nbytes, what, key_name or name))
        if flops_executed is not None:
            executed[(key_name or name, P)] = flops_executed

    # bytes of a buffer's 16 never-touched slab slots per 32-point tile (the buffers keep the slots: csrc/mlp_layout.h kActFeat / kDyFeat)
    slab_b = {"fp32": 2048, "bf16": 1024, "bf16_f8": 512}[dtype]

    def unsaved(P):
        ppw = 128 if dtype == "fp32" else 256
        return (P + ppw - 1) // ppw * (ppw // 32) * UNSAVED_SLABS_PER_TILE * slab_b

    keep, entries, dw_b, P_all, chain_b = [], [], 0, 0, 0
    one_fwd = merged and ops.render_supported(B, S, N, dtype)       # the step's forward is ONE launch (nerfhip_render_train_fwd)
    from nerf_pl_amd.models.train_step import regen_enc_enabled
    # bf16 step: the 6 input-encoding slabs of a tile are not saved; the dW launch forms the 10 it would read from (rays, z)
    regen = one_fwd and regen_enc_enabled() and dtype == "bf16" and S % 32 == 0 and (S + N) % 32 == 0
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
        act_b, dy_b = acts.numel() - unsaved(P), ws["dys"].numel() - unsaved(P)        # bytes written / read, not allocated
        gate_b = (P + 31) // 32 * 9 * 1024
        enc_w = (P + 31) // 32 * 6 * slab_b if regen else 0
        # split-K partials the reduce kernel reads: per split 528 used (out-tile, x-tile) blocks of 4 KiB over the 10 jobs with workgroups
        ws_b = int(lib.nerfhip_mlp_dw_splits(P, code)) // 10 * 528 * 4096
        if not one_fwd:
            entry("mlp_fwd_kernel<save>", tag, P, lambda zz=zz, pk=pk, acts=acts: ops.mlp_fwd_rays(rays, zz, pk, False, dtype, save=acts),
                  FLOP_PER_POINT_FULL * P, act_b + 20 * P, "saved activations + gates written once, 4 B z in + 16 B out per point",
                  flops_executed=FLOP_PER_POINT_FULL_EXECUTED * P)
        fwd_bytes += act_b - enc_w + 56 * P         # + per point: 16 B raw written and read back twice by the compositing waves, 16 B d loss / d raw, z
        if not merged:
            entry("mlp_bwd_chain_kernel", tag, P,
                  lambda g_out=g_out, raw=raw, pb=pb, acts=acts, ws=ws: ops.mlp_bwd(g_out, raw, pb, acts, dtype, phases=1, workspace=ws),
                  FLOP_PER_POINT_DX * P, dy_b + gate_b + 32 * P, "dY written once, ReLU gate words + g_out/out read",
                  flops_executed=FLOP_PER_POINT_DX_EXECUTED * P)
        chain_b += dy_b + gate_b + 32 * P
        if not merged:
            entry("mlp_bwd_dw_kernel", tag, P,
                  lambda g_out=g_out, raw=raw, pb=pb, acts=acts, ws=ws: ops.mlp_bwd(g_out, raw, pb, acts, dtype, phases=2, workspace=ws),
                  FLOP_PER_POINT_DW * P, (act_b - gate_b) + dy_b, "every saved activation and dY slab read once",
                  flops_executed=FLOP_PER_POINT_DW_EXECUTED * P)
            entry("mlp_bwd_reduce_kernel", tag, P,
                  lambda g_out=g_out, raw=raw, pb=pb, acts=acts, ws=ws: ops.mlp_bwd(g_out, raw, pb, acts, dtype, phases=4, workspace=ws),
                  0, ws_b + 2 * 595844 * 4, "split-K partial slabs read, 24 gradient tensors written")
        entries.append((g_out, raw, pb, acts) + ((rays, zz) if regen else ()))
        keep.append((acts, raw, g_out, ws))
        dw_b += (act_b - gate_b) + dy_b - enc_w      # (regen: the encoding slabs are neither saved nor read)
        P_all += P
    if one_fwd:
        tgt_ = torch.rand(B, 3, device=dev)
        pr_, u_ = torch.rand(B, S, device=dev), torch.rand(B, N, device=dev)
        gs_ = 2.0 / (3 * B)
        a_c, a_f = keep[1][0], keep[0][0]
        pk_c, pk_f = models[0].packed_weights(dtype), models[1].packed_weights(dtype)
        entry("mlp_render_kernel<train>", "the step's whole forward in ONE launch: coarse + fine MLP, compositing, loss gradient, fine depths, loss",
              P_all, lambda: ops.render_train_fwd(rays, tgt_, gs_, S, N, pk_c, pk_f, dtype, a_c, a_f, False, 1.0, pr_, None, None, 0.0, True, u_, regen_enc=regen),
              FLOP_PER_POINT_FULL * P_all, fwd_bytes,
              "saved activations + gates of both models written once; per point 16 B rgb sigma out and back (L2), 16 B d loss / d raw, depths",
              key_name="mlp_render_kernel<train>", flops_executed=FLOP_PER_POINT_FULL_EXECUTED * P_all)
    if merged:
        wsm = {}
        ops.mlp_bwd_multi(entries, dtype, workspace=wsm)            # (chains included: fills the dY slabs the dW launch reads)
        n_arr = (__import__("ctypes").c_int64 * 2)(*[e[1].numel() // 4 for e in entries])
        n_slabs = int(lib.nerfhip_mlp_dw_workspace_bytes_multi(n_arr, 2, code)) // (4 * (8 * 10 * 64 * 16 + 8 * 64))
        ws_b = n_slabs * 528 * 4096 // 10        # average used blocks per partial slab (528 of a model's 10 jobs with workgroups together)
        entry("mlp_bwd_chain_kernel", "fine + coarse pass in ONE launch", P_all, lambda: ops.mlp_bwd_multi(entries, dtype, phases=1, workspace=wsm),
              FLOP_PER_POINT_DX * P_all, chain_b, "dY of both models written once, ReLU gate words + g_out/out read", key_name="mlp_bwd_chain_kernel<merged>",
              flops_executed=FLOP_PER_POINT_DX_EXECUTED * P_all)
        entry("mlp_bwd_dw_kernel", "fine + coarse pass in ONE launch", P_all, lambda: ops.mlp_bwd_multi(entries, dtype, phases=2, workspace=wsm),
              FLOP_PER_POINT_DW * P_all, dw_b, "every saved activation and dY slab of both models read once", key_name="mlp_bwd_dw_kernel<merged>",
              flops_executed=FLOP_PER_POINT_DW_EXECUTED * P_all)
        entry("mlp_bwd_reduce_kernel", "both models in ONE launch, + mlp_bwd_fold_kernel behind it", P_all, lambda: ops.mlp_bwd_multi(entries, dtype, phases=4, workspace=wsm),
              0, ws_b + 4 * 595844 * 4, "split-K partial slabs read, 48 gradient tensors written (6 of them by the fold launch)", key_name="mlp_bwd_reduce_kernel<merged>")
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
        if (key_name, P) in executed:
            fe = executed[(key_name, P)]
            out[-1].update({"flops_executed": fe, "frac_mfma_executed_in_step": round(fe / max(in_mix[k], 1e-3) / 1e6 / peak, 4),
                            "flops_note": "`flops` = the reference's GEMMs (SURVEY 8d); the launch executes `flops_executed`: xyz_encoding_final "
                                          "(no activation) is folded into the dir layer — one 256 x 256 product per point fewer"})
    out.sort(key=lambda r: -r["avg_launch_us"])
    del keep, entries, todo
    return out, round(mix_us, 1)


def side_steps(a, hp, build_system, make_stepper, timed, dev):
    """Extras of the default line, measured AFTER the headline's timed region: the fp8-dW variant of the same step (its own
    process: a dedicated run, not a second system squeezed into this one), BASELINE configs[1] (fp32, 64+64) and configs[3]
    (NDC rays, noise_std=1, black background, 64+64) training steps."""
    import subprocess
    from argparse import Namespace

    from bench_inputs import DTYPE_LABEL, synth_store_ndc
    B, S, N = a.rays, a.n_samples, a.n_importance
    ex = {}
    step_flops_pt = FLOP_PER_POINT_FULL + FLOP_PER_POINT_DX + FLOP_PER_POINT_DW
    if a.dtype == "bf16":
        cmd = [sys.executable, BENCH_PY, "--dtype", "bf16_f8", "--steps", "15", "--warmup", "6", "--no-extras",
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
    for _ in range(4 + 12):                                # build (3 eager steps + capture) + a short settle: the same protocol as the
        st1()                                              # headline's `value` (sustained), scaled to this 8 ms step
    t1 = timed(st1, 5, 8) / 8
    ex["fp32_c1_ms_per_step"] = round(t1 * 1e3, 4)
    ex["fp32_c1_frac_mfma"] = round(step_flops_pt * B * (2 * S + 64) / t1 / 1e12 / PEAK_TFLOPS["fp32"], 4)
    ex["fp32_c1_note"] = "configs[1]: %d rays x (%d+64) samples, exact-fp32 MFMA MLP, full training step; frac of the 157.3 TFLOP/s fp32 MFMA peak" % (B, S)
    del sys1, opt1, st1
    # configs[3]: LLFF-style NDC rays (non-unit directions), noise_std = 1 (rendering.py:152 noise path), black background, 64 + 64
    hp3 = Namespace(**dict(vars(hp), N_importance=64, noise_std=1.0, white_back=False))
    sys3, opt3 = build_system(a.dtype, hp3)
    st3, _ = make_stepper(sys3, opt3, None, synth_store_ndc(777, dev))
    # Same protocol as the headline's `value`: build, then settle replays, then W + K.  (Rounds 4-5 timed 15 replays right after the
    # capture — the device's cold state — and reported 0.966 ms = 0.284 of the peak for a step that sustains 0.81 ms = 0.339:
    # `python bench.py --workload c3`, profiles/r06_ndc_c3_*.)
    for _ in range(4 + 120):
        st3()
    t3 = timed(st3, 5, 15) / 15
    ex["ndc_c3_ms_per_step"] = round(t3 * 1e3, 4)
    ex["ndc_c3_frac_mfma"] = round(step_flops_pt * B * (2 * S + 64) / t3 / 1e12 / PEAK_TFLOPS[a.dtype], 4)
    ex["ndc_c3_note"] = ("configs[3] per GPU: %d NDC rays x (%d+64) samples, noise_std=1, white_back=False, %s, full training step, sustained "
                         "(120 settle replays, then 5 + 15; `python bench.py --workload c3` is the same step as a line of its own; the 8-GPU "
                         "half of configs[3] is the --gpus N line)" % (B, S, DTYPE_LABEL[a.dtype]))
    del sys3, opt3, st3
    return ex
