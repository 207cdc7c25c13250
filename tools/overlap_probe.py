"""Probe (run on the GPU box): does the backward of the training step gain from running the HBM-read-bound weight-gradient launch of
one model NEXT TO the write-heavy chain launch of the other (two streams inside one hipGraph)?
    python tools/overlap_probe.py [--dtype bf16]
Sequences, all over the same resident saved tensors of 1024 x (64 + 128):
  merged     chain(both) -> dW(both) -> reduce(both)                    (what the step runs: nerfhip_mlp_bwd_multi)
  serial     chain(c) -> chain(f) -> dW(c) -> dW(f) -> reduce x 2       (per-model launches, one stream)
  overlap_c  chain(c) -> [ chain(f) || dW(c) ] -> dW(f) -> reduce x 2
  overlap_f  chain(f) -> [ chain(c) || dW(f) ] -> dW(c) -> reduce x 2
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import bench_extras  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    from nerf_pl_amd import ops
    from nerf_pl_amd.models.nerf import NeRF
    dev = torch.device("cuda", 0)
    B, S, N = 1024, 64, 128
    dtype = a.dtype
    models = []
    for seed in (100, 101):
        m = NeRF()
        m.load_state_dict(bench.synth_params(seed, 4.0, 0.2))
        m.mlp_dtype = dtype
        models.append(m.to(dev))
    rays = bench.synth_rays(1234, B).to(dev)
    with torch.no_grad():
        z = ops.sample_coarse_z(rays, S, False, 0.0)
        zf = ops.fine_z(z, torch.rand(B, S, device=dev), N)
    per = {}
    entries = []
    for tag, model, zz in (("f", models[1], zf), ("c", models[0], z)):
        P = zz.numel()
        pk, pb = model.packed_weights(dtype), model.packed_weights_bwd(dtype)
        acts = ops.alloc_acts(P, dtype, dev)
        raw = ops.mlp_fwd_rays(rays, zz, pk, False, dtype, save=acts)
        g_out = torch.randn_like(raw)
        ws = {}
        ops.mlp_bwd(g_out, raw, pb, acts, dtype, workspace=ws)
        per[tag] = (g_out, raw, pb, acts, ws)
        entries.append((g_out, raw, pb, acts))
    wsm = {}
    ops.mlp_bwd_multi(entries, dtype, workspace=wsm)

    def ph(tag, phases):
        g_out, raw, pb, acts, ws = per[tag]
        ops.mlp_bwd(g_out, raw, pb, acts, dtype, phases=phases, workspace=ws)

    side = torch.cuda.Stream()

    def merged():
        ops.mlp_bwd_multi(entries, dtype, phases=7, workspace=wsm)

    def serial():
        ph("c", 1); ph("f", 1); ph("c", 2); ph("f", 2); ph("c", 4); ph("f", 4)

    def overlap(first, second):
        def fn():
            cur = torch.cuda.current_stream()
            ph(first, 1)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ph(first, 2)
            ph(second, 1)
            cur.wait_stream(side)
            ph(second, 2)
            ph(first, 4); ph(second, 4)
        return fn

    for name, fn in (("merged", merged), ("serial", serial), ("overlap_c", overlap("c", "f")), ("overlap_f", overlap("f", "c")),
                     ("merged", merged), ("overlap_c", overlap("c", "f"))):
        avg, mn = bench_extras.event_time(fn, 6, graph=True)
        print("%-10s %8.1f us avg  %8.1f us min" % (name, avg, mn), flush=True)


if __name__ == "__main__":
    main()
