"""HBM-bound kernels of the path at eval-chunk size (32768 rays x 64/192 samples): achieved GB/s on the algorithmic
bytes of SURVEY §8(d) against the 8 TB/s HBM3E peak.  HIP events, random data, 20 reps each."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _synth  # noqa: E402  (tools/_synth.py: seeded synthetic rays / poses)
from nerf_pl_amd import ops, rays as R  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return sum(ev[i].elapsed_time(ev[i + 1]) for i in range(reps)) / reps * 1e3     # us


def main():
    dev = torch.device("cuda:0")
    B, Sc, Ni = 32768, 64, 128
    Sf = Sc + Ni
    rays = _synth.make_rays(1, B, dev)
    rows = []

    def rec(name, us, nbytes, note):
        rows.append({"kernel": name, "us": round(us, 1), "algorithmic_MB": round(nbytes / 1e6, 1),
                     "GB_per_s": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 3), "shape": note})

    n = B * Sc
    x = torch.rand(n, 3, device=dev) * 8 - 4
    rec("posenc (3->63)", timed(lambda: ops.posenc(x, 10)), n * 264, "%d points" % n)
    z = ops.sample_coarse_z(rays, Sc, False, 0.0)
    pr = torch.rand(B, Sc, device=dev)
    rec("sample_coarse_z (perturb)", timed(lambda: ops.sample_coarse_z(rays, Sc, False, 1.0, pr)), B * (32 + 8 * Sc), "%d rays x %d" % (B, Sc))
    sig = torch.randn(B, Sc, device=dev) * 3
    w, _ = ops.composite(sig, z, rays, None, 0.0, True)
    rec("composite_fwd sigma-only S=64", timed(lambda: ops.composite(sig, z, rays, None, 0.0, True)), B * (Sc * 12 + 12 + 4), "%d rays" % B)
    u = torch.rand(B, Ni, device=dev)
    rec("fine_z (sample_pdf + merge), u given", timed(lambda: ops.fine_z(z, w, Ni, u=u)), B * 4 * (2 * Sc + 2 * Ni + Sf), "%d rays, 64+128" % B)
    rec("fine_z deterministic", timed(lambda: ops.fine_z(z, w, Ni, u=None)), B * 4 * (2 * Sc + Sf), "%d rays, 64+128" % B)
    bins = 0.5 * (z[:, 1:] + z[:, :-1])
    rec("sample_pdf (N_i=128, det)", timed(lambda: ops.sample_pdf_u(bins, w[:, 1:-1], Ni)), B * 1012, "%d rays" % B)
    cdf = torch.sort(torch.rand(B, 63, device=dev), -1)[0]
    rec("searchsorted_right (63 x 128)", timed(lambda: ops.searchsorted(cdf, u, side="right")), B * (63 * 4 + Ni * 4 + Ni * 8), "%d rows" % B)
    zf = ops.fine_z(z, w, Ni, u=u)
    raw = torch.randn(B, Sf, 4, device=dev)
    rec("composite_fwd S=192", timed(lambda: ops.composite(raw, zf, rays, None, 0.0, True)), B * 4640, "%d rays" % B)
    raw_g = raw.clone().requires_grad_(True)
    wts, opac, rgb, dep = ops.composite(raw_g, zf, rays, None, 0.0, True)
    g = torch.randn_like(rgb)
    rec("composite_bwd S=192 (g_rgb only)", timed(lambda: torch.autograd.grad(rgb, raw_g, g, retain_graph=True)), B * (Sf * 20 + 12 + Sf * 16), "%d rays" % B)
    poses = torch.stack([_synth.make_pose(i) for i in range(100)]).to(dev)
    ids = torch.randint(0, 100 * 800 * 800, (1 << 20,), device=dev)
    rec("gen_rays (pixel ids -> rays)", timed(lambda: R.gen_rays(poses, 800, 800, 1111.1, 2.0, 6.0, pixel_ids=ids)), (1 << 20) * 40, "1,048,576 rays")
    out = os.path.join(ROOT, "gpurun_out", "kbench_hbm.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rows, open(out, "w"), indent=1)
    for r in rows:
        print("%-40s %8.1f us  %8.1f MB  %7.1f GB/s  %5.1f %% of 8 TB/s   %s" % (r["kernel"], r["us"], r["algorithmic_MB"], r["GB_per_s"],
                                                                              100 * r["frac_of_8TBps"], r["shape"]))


if __name__ == "__main__":
    main()
