"""Per-kernel micro-benchmark of the MLP kernels (HIP events, same process) for A/B-ing library builds:

    NERFHIP_LIB_PATH=nerf_pl_amd/variants/libnerfhip_X.so python tools/kbench.py [--rays 1024] [--samples 192] [--dtype bf16]

Prints one line: fwd (inference), fwd+save, bwd (chain+dW+reduce) average microseconds over `--reps` launches
on random (not zero) data.  Individual backward kernels are split out by `rocprofv3 --kernel-trace`."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _synth  # noqa: E402  (tools/_synth.py: seeded synthetic rays / weights)
from nerf_pl_amd import ops  # noqa: E402
from nerf_pl_amd.models import NeRF  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return sum(ts) / len(ts), ts[0]


def merged(a, dev, m_fine):
    B = a.rays
    rays = _synth.make_rays(1, B, dev)
    m_coarse = _synth.make_model(100, dev, a.dtype)
    entries = []
    for m, S in ((m_fine, a.samples), (m_coarse, 64)):
        z = torch.sort(2 + 4 * torch.rand(B, S, device=dev), -1)[0]
        acts = ops.alloc_acts(B * S, a.dtype, dev)
        pk, pb = m.packed_weights_train(a.dtype)
        out = ops.mlp_fwd_rays(rays, z, pk, False, a.dtype, save=acts)
        entries.append((torch.randn_like(out), out, pb, acts))
    ws = {}
    ops.mlp_bwd_multi(entries, a.dtype, workspace=ws)                       # chains: fill the dY slabs
    d_avg, d_min = timed(lambda: ops.mlp_bwd_multi(entries, a.dtype, phases=2, workspace=ws), a.reps)
    r_avg, r_min = timed(lambda: ops.mlp_bwd_multi(entries, a.dtype, phases=4, workspace=ws), a.reps)
    print("%s merged %dx(%d+64) %s: dW %.1f (min %.1f)  reduce %.1f (min %.1f) us"
          % (os.path.basename(os.environ.get("NERFHIP_LIB_PATH", "libnerfhip.so")), B, a.samples, a.dtype, d_avg, d_min, r_avg, r_min), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--interleaved", action="store_true")
    ap.add_argument("--merged", action="store_true",
                    help="only the launches the fused training step shares between its two models: ONE dW launch + ONE reduce launch over "
                         "a fine pass of --samples and a coarse pass of 64 samples per ray")
    ap.add_argument("--clock-probe", action="store_true",
                    help="library built with -DNERFHIP_CLOCK_PROBE=1, --dtype bf16_f8: shader clock and cycles of the saving forward")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = _synth.make_model(101, dev, a.dtype)
    B, S = a.rays, a.samples
    if a.merged:
        return merged(a, dev, m)
    rays = _synth.make_rays(1, B, dev)
    z = torch.sort(2 + 4 * torch.rand(B, S, device=dev), -1)[0]
    packed = m.packed_weights(a.dtype)
    pb = m.packed_weights_bwd(a.dtype)
    acts = ops.alloc_acts(B * S, a.dtype, dev)
    out = ops.mlp_fwd_rays(rays, z, packed, False, a.dtype, save=acts)
    g_out = torch.randn_like(out)
    f_avg, f_min = timed(lambda: ops.mlp_fwd_rays(rays, z, packed, False, a.dtype), a.reps)
    s_avg, s_min = timed(lambda: ops.mlp_fwd_rays(rays, z, packed, False, a.dtype, save=acts), a.reps)
    ws = {}
    b_avg, b_min = timed(lambda: ops.mlp_bwd(g_out, out, pb, acts, a.dtype, workspace=ws), a.reps)
    ph = [timed(lambda ph=ph: ops.mlp_bwd(g_out, out, pb, acts, a.dtype, phases=ph, workspace=ws), a.reps) for ph in (1, 2, 4)]
    so_avg, so_min = timed(lambda: ops.mlp_fwd_rays(rays, z, packed, True, a.dtype), a.reps)
    if a.interleaved:
        # the training step's order: [pack, fwd+save, chain, dW, reduce] repeated, every kernel bracketed by events
        names = ["pack", "fwd+save", "chain", "dW", "reduce"]
        fns = [lambda: m.packed_weights(a.dtype),
               lambda: ops.mlp_fwd_rays(rays, z, packed, False, a.dtype, save=acts),
               lambda: ops.mlp_bwd(g_out, out, pb, acts, a.dtype, phases=1, workspace=ws),
               lambda: ops.mlp_bwd(g_out, out, pb, acts, a.dtype, phases=2, workspace=ws),
               lambda: ops.mlp_bwd(g_out, out, pb, acts, a.dtype, phases=4, workspace=ws)]
        tot = [0.0] * len(fns)
        for rep in range(a.reps + 3):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(fns) + 1)]
            evs[0].record()
            for i, f in enumerate(fns):
                f()
                evs[i + 1].record()
            torch.cuda.synchronize()
            if rep >= 3:
                for i in range(len(fns)):
                    tot[i] += evs[i].elapsed_time(evs[i + 1]) * 1e3
        print("   interleaved (step order): " + "  ".join("%s %.1f" % (n, t / a.reps) for n, t in zip(names, tot)), flush=True)
    if a.clock_probe and a.dtype == "bf16_f8":
        for _ in range(10):
            ops.mlp_fwd_rays(rays, z, packed, False, a.dtype, save=acts)
        torch.cuda.synchronize()
        tiles = ((B * S + 255) // 256) * 8
        tb = acts.numel() // tiles
        pr = acts.view(tiles, tb)[:, (79 + 9) * 1024 + 64:(79 + 9) * 1024 + 72].contiguous().view(torch.int32).double()
        cyc, wall = pr[:, 0], pr[:, 1]
        print("   clock probe (fwd+save): %.0f kcycles/wave (min %.0f max %.0f), %.1f us/wave, shader clock %.0f MHz"
              % (cyc.mean().item() / 1e3, cyc.min().item() / 1e3, cyc.max().item() / 1e3, wall.mean().item() / 100.0,
                 (cyc.sum() / wall.sum()).item() * 100.0), flush=True)
    print("%s %dx%d %s: fwd %.1f (min %.1f)  fwd_sigma %.1f  fwd+save %.1f (min %.1f)  bwd %.1f (min %.1f) = chain %.1f + dW %.1f + reduce %.1f us"
          % (os.path.basename(os.environ.get("NERFHIP_LIB_PATH", "libnerfhip.so")), B, S, a.dtype, f_avg, f_min, so_avg, s_avg, s_min,
             b_avg, b_min, ph[0][0], ph[1][0], ph[2][0]), flush=True)


if __name__ == "__main__":
    main()
