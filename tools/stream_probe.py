"""How long do the waves of the activation-saving forward and of the backward chain sit at the weight ring's boundaries?
Needs a library built with -DNERFHIP_STREAM_PROBE=1 (results of those launches are invalid):

    NERFHIP_BUILD_TAG=stprobe NERFHIP_EXTRA_FLAGS=-DNERFHIP_STREAM_PROBE=1 python -m nerf_pl_amd.build
    NERFHIP_LIB_PATH=nerf_pl_amd/variants/libnerfhip_stprobe.so python tools/stream_probe.py [--dtype bf16]

Every wave records, in 10 ns ticks of s_memrealtime: the time at the boundaries' `s_waitcnt vmcnt(n)` (its weight DMAs of the chunk —
and, vmcnt retiring in order, every activation / dY store older than them — have completed), the time at the `s_barrier` behind it
(the other waves), the boundaries it passed and its wall time."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _synth  # noqa: E402
from nerf_pl_amd import ops  # noqa: E402


def report(name, raw, tiles, tile_bytes):
    pr = raw.view(tiles, tile_bytes)[:, :16].contiguous().view(torch.int32).double().cpu()
    wait, bar, n, wall = pr[:, 0], pr[:, 1], pr[:, 2], pr[:, 3]
    print("%-22s %6d waves  boundaries %3.0f | per wave: wall %6.1f us (max %6.1f)  at s_waitcnt %5.1f us (%4.1f %%, max %5.1f)  at s_barrier %5.1f us (%4.1f %%)"
          % (name, tiles, n.mean(), wall.mean() / 100, wall.max() / 100, wait.mean() / 100, 100 * wait.sum() / wall.sum(), wait.max() / 100,
             bar.mean() / 100, 100 * bar.sum() / wall.sum()), flush=True)
    # is the barrier time systematic?  mean per wave slot of the workgroup (tile = workgroup * 8 + wave), and how much of a
    # wave's barrier time is explained by its slot (between-slot variance / total variance)
    b8, w8 = bar.view(-1, 8) / 100, wait.view(-1, 8) / 100
    print("      at s_barrier by wave slot (us): " + " ".join("%5.1f" % v for v in b8.mean(0).tolist())
          + "   | spread within a slot (std) %.1f us, between slots %.1f us" % (b8.std(0).mean().item(), b8.mean(0).std().item()), flush=True)
    print("      the workgroup's fastest / slowest wave at its barriers: %.1f / %.1f us"
          % (b8.min(1)[0].mean().item(), b8.max(1)[0].mean().item()), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rays", type=int, default=1024)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.rays
    rays = _synth.make_rays(1, B, dev)
    print("%s  %s" % (os.path.basename(os.environ.get("NERFHIP_LIB_PATH", "libnerfhip.so")), a.dtype))
    for S in (192, 64):
        m = _synth.make_model(101, dev, a.dtype)
        z = torch.sort(2 + 4 * torch.rand(B, S, device=dev), -1)[0]
        acts = ops.alloc_acts(B * S, a.dtype, dev)
        pk, pb = m.packed_weights_train(a.dtype)
        tiles = (B * S + 255) // 256 * 8
        for _ in range(3):
            out = ops.mlp_fwd_rays(rays, z, pk, False, a.dtype, save=acts)
        torch.cuda.synchronize()
        report("fwd+save %dx%d" % (B, S), acts, tiles, acts.numel() // tiles)
        ws = {}
        g = torch.randn_like(out)
        for _ in range(3):
            ops.mlp_bwd_multi([(g, out, pb, acts)], a.dtype, phases=1, workspace=ws)
        torch.cuda.synchronize()
        (dys, _), = ws.values()                     # ops.mlp_bwd_multi keeps (dY tensors per model, dW workspace) per shape key
        d = dys[0]
        report("chain %dx%d" % (B, S), d, tiles, d.numel() // tiles)


if __name__ == "__main__":
    main()
