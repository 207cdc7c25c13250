"""Where do the dW kernel's cycles go?  Needs a library built with -DNERFHIP_DW_PROBE=1:

    NERFHIP_BUILD_TAG=dwprobe NERFHIP_EXTRA_FLAGS=-DNERFHIP_DW_PROBE=1 python -m nerf_pl_amd.build
    NERFHIP_LIB_PATH=nerf_pl_amd/variants/libnerfhip_dwprobe.so python tools/dw_probe.py [--dtype bf16]

Runs the training step's merged weight-gradient launch (fine 1024 x 192 + coarse 1024 x 64 points) and prints, per job, the mean
over its workgroups' waves of: ring iterations, cycles per iteration spent waiting for the stage's DMAs (s_waitcnt vmcnt), at the
barrier, issuing the next stage, in the MFMA / LDS-read block; ring depth; wall time of the workgroup (100 MHz ticks -> us)."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _synth  # noqa: E402
from nerf_pl_amd import _lib, ops  # noqa: E402

JOBS = ["first", "L2", "L3", "L4", "skip", "L6", "L7", "L8", "final", "dir", "sigma", "rgb"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rays", type=int, default=1024)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.rays
    rays = _synth.make_rays(1, B, dev)
    entries = []
    for seed, S in ((101, 192), (100, 64)):
        m = _synth.make_model(seed, dev, a.dtype)
        z = torch.sort(2 + 4 * torch.rand(B, S, device=dev), -1)[0]
        acts = ops.alloc_acts(B * S, a.dtype, dev)
        pk, pb = m.packed_weights_train(a.dtype)
        out = ops.mlp_fwd_rays(rays, z, pk, False, a.dtype, save=acts)
        entries.append((torch.randn_like(out), out, pb, acts))
    ws = {}
    ops.mlp_bwd_multi(entries, a.dtype, workspace=ws)
    for _ in range(5):
        ops.mlp_bwd_multi(entries, a.dtype, phases=2, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.mlp_bwd_multi(entries, a.dtype, phases=2, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    lib = _lib.load()
    fn = lib.nerfhip_debug_dw_probe
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    buf = np.zeros(1024 * 8 * 8, dtype=np.uint32)
    rc = fn(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    pr = buf.reshape(1024, 8, 8).astype(np.float64)
    used = pr[:, 0, 0] > 0
    print("%s  dW launch %.1f us, %d workgroups" % (os.path.basename(os.environ.get("NERFHIP_LIB_PATH", "libnerfhip.so")),
                                                    e0.elapsed_time(e1) * 1e3, int(used.sum())))
    print("job        wgs  depth  iters | 10ns ticks per iteration: wait  barrier  issue  compute  sum | wg wall us (mean, max)")
    for j in range(24):
        sel = used & (pr[:, 0, 6] == j)
        if not sel.any():
            continue
        w = pr[sel]
        it = w[:, :, 0].mean()
        per = [w[:, :, k].sum() / w[:, :, 0].sum() for k in (1, 2, 3, 4)]
        wall = w[:, :, 5].max(axis=1) / 100.0
        print("%-6s m%d  %3d  %5d  %5d |            %6.0f  %6.0f  %6.0f  %6.0f  %6.0f | %6.1f %6.1f"
              % (JOBS[j % 12], j // 12, int(sel.sum()), int(w[0, 0, 7]), it, per[0], per[1], per[2], per[3], sum(per), wall.mean(), wall.max()))


if __name__ == "__main__":
    main()
