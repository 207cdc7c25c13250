"""A/B of library builds at the tensor level: run the saving forward + backward on fixed seeded inputs with the library in
NERFHIP_LIB_PATH and dump dY slabs + gradients; `--compare a b` reports where two dumps differ.
    NERFHIP_LIB_PATH=... python tests/tools/dbg_chain_ab.py --dump gpurun_out/x/dump_tag.pt
    python tests/tools/dbg_chain_ab.py --compare gpurun_out/x/dump_a.pt gpurun_out/x/dump_b.pt"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def dump(path):
    from oracle import nerf_oracle as O
    from nerf_pl_amd import ops
    from tests.helpers import build_models
    dev = torch.device("cuda:0")
    out = {}
    for dtype in ("bf16", "bf16_f8", "fp32"):
        for n in (1, 1000):
            g = torch.Generator().manual_seed(n)
            p = O.make_params(21, 3.0, 0.1)
            pts = torch.rand(n, 3, generator=g) * 4 - 2
            dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
            x = torch.cat([O.posenc(pts, 10), O.posenc(dirs, 4)], 1).to(dev)
            g_out = torch.randn(n, 4, generator=g).to(dev)
            (m,), _ = build_models([p], dev, dtype)
            acts = ops.alloc_acts(n, dtype, dev)
            acts.zero_()
            packed, pb = m.packed_weights_train(dtype)
            o = ops.mlp_fwd_embedded(x, packed, False, dtype, save=acts)
            ws = {}
            gw, gb, flat = ops.mlp_bwd(g_out, o, pb, acts, dtype, workspace=ws)
            torch.cuda.synchronize()
            out[(dtype, n)] = {"out": o.cpu(), "acts": acts.cpu(), "dys": ws["dys"].cpu(), "flat": flat.cpu()}
    torch.save(out, path)
    print("dumped", path)


def compare(a, b):
    A, B = torch.load(a), torch.load(b)
    for k in A:
        for name in ("out", "acts", "dys", "flat"):
            x, y = A[k][name], B[k][name]
            if x.dtype == torch.uint8:
                diff = (x != y)
                print(k, name, "bytes differing: %d of %d" % (diff.sum().item(), x.numel()),
                      ("first at %d" % diff.nonzero()[0].item()) if diff.any() else "")
            else:
                d = (x.float() - y.float()).abs()
                print(k, name, "max abs diff %.3e (max |a| %.3e), nan a/b %d/%d" % (d.max().item(), x.abs().max().item(),
                                                                                       torch.isnan(x).sum().item(), torch.isnan(y).sum().item()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dump")
    ap.add_argument("--compare", nargs=2)
    a = ap.parse_args()
    if a.dump:
        dump(a.dump)
    else:
        compare(*a.compare)
