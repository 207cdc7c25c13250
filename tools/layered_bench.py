"""Timing of the layer-by-layer path (csrc/linear.hip) next to the fused kernels on the DEFAULT shape, and of one non-default
shape: forward and forward+backward of NeRF.forward on n pre-embedded points.   python tools/layered_bench.py [--n 196608]"""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_pl_amd.models import NeRF  # noqa: E402
from nerf_pl_amd.models.layered import nerf_forward  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=196608)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for dtype in ("bf16", "fp32"):
        for tag, kw in (("default D8 W256", {}), ("D4 W128 39/15", dict(D=4, W=128, in_channels_xyz=39, in_channels_dir=15, skips=[2]))):
            torch.manual_seed(0)
            m = NeRF(**kw).to(dev)
            m.mlp_dtype = dtype
            x = torch.rand(a.n, m.in_channels_xyz + m.in_channels_dir, device=dev) * 2 - 1
            g = torch.randn(a.n, 4, device=dev)
            macs = sum(p.numel() for n_, p in m.named_parameters() if n_.endswith("weight"))

            def fwd(layered):
                with torch.no_grad():
                    return nerf_forward(m, x) if layered else m(x)

            def fwdbwd(layered):
                m.zero_grad(set_to_none=True)
                out = nerf_forward(m, x) if layered else m(x)
                out.backward(g)

            row = {}
            for layered in ([True, False] if m.is_default_arch() else [True]):
                f = timed(lambda: fwd(layered), a.reps)
                fb = timed(lambda: fwdbwd(layered), a.reps)
                row["layered" if layered else "fused"] = (f, fb)
            peak = 2500.0 if dtype == "bf16" else 157.3
            msg = "%-16s %-5s n=%d" % (tag, dtype, a.n)
            for k, (f, fb) in row.items():
                msg += " | %s fwd %.0f us (%.3f of peak) fwd+bwd %.0f us" % (k, f, 2 * macs * a.n / (f * 1e-6) / 1e12 / peak, fb)
            print(msg, flush=True)


if __name__ == "__main__":
    main()
