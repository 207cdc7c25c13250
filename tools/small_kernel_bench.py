"""Timing of the step's small kernels alone, each replayed from a hipGraph (20 launches per replay): fine_z, composite_train,
sample_coarse_z at the benchmark shape (1024 rays, 64 + 128 samples).   python tools/small_kernel_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_pl_amd import ops  # noqa: E402


def graph_time(fn, n=20, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, S, N = 1024, 64, 128
    rays = torch.cat([torch.randn(B, 3) * 0.1 + torch.tensor([0, 0, 4.0]), torch.nn.functional.normalize(torch.randn(B, 3), dim=-1),
                      torch.full((B, 1), 2.0), torch.full((B, 1), 6.0)], 1).to(dev)
    pr = torch.rand(B, S, device=dev)
    z = ops.sample_coarse_z(rays, S, False, 1.0, pr)
    w = torch.rand(B, S, device=dev) ** 4
    u = torch.rand(B, N, device=dev)
    raw = torch.randn(B, S + N, 4, device=dev)
    zf = ops.fine_z(z, w, N, u=u)
    noise = torch.randn(B, S + N, device=dev)
    tgt = torch.rand(B, 3, device=dev)
    print("lib %s" % os.path.basename(os.environ.get("NERFHIP_LIB_PATH", "libnerfhip.so")),
          "fine_z %.2f us" % graph_time(lambda: ops.fine_z(z, w, N, u=u)),
          "fine_z(det) %.2f us" % graph_time(lambda: ops.fine_z(z, w, N, u=None)),
          "sample_coarse_z %.2f us" % graph_time(lambda: ops.sample_coarse_z(rays, S, False, 1.0, pr)),
          "composite_train(192) %.2f us" % graph_time(lambda: ops.composite_train(raw, zf, rays, noise, 1.0, True, tgt, 1e-3, want_weights=False)),
          flush=True)
    # round 4: the fused launches next to what they replace
    from nerf_pl_amd import draws as D
    from nerf_pl_amd.rays import RayStore
    raw_c = torch.randn(B, S, 4, device=dev)
    rgb_c = torch.rand(B, 3, device=dev)
    st = RayStore(torch.eye(3, 4).repeat(20, 1, 1).to(dev), torch.rand(20 * 200 * 200, 3, device=dev), 200, 200, 300.0, 2.0, 6.0)
    ds = D.GraphDrawState(dev)
    ds.arm()
    with D.capturing(ds):
        print("composite_train(64) %.2f us" % graph_time(lambda: ops.composite_train(raw_c, z, rays, None, 0.0, True, tgt, 1e-3, want_weights=True)),
              "composite_train_fine_z(64,128) %.2f us" % graph_time(lambda: ops.composite_train_fine_z(raw_c, z, rays, None, 0.0, True, tgt, 1e-3, N, u=u)),
              "composite_train_loss(192) %.2f us" % graph_time(lambda: ops.composite_train_loss(raw, zf, rays, None, 0.0, True, tgt, 1e-3, rgb_coarse=rgb_c)),
              "mse_psnr %.2f us" % graph_time(lambda: ops.mse_psnr_values(rgb_c, rgb_c, tgt)),
              "batch+draws %.2f us" % graph_time(lambda: st.sample(B, step_draws=(S, N, 1.0, 0.0))),
              "batch only %.2f us" % graph_time(lambda: st.sample(B)),
              "draws only %.2f us" % graph_time(lambda: D.step_draws(B, S, N, 1.0, 0.0, dev)), flush=True)


if __name__ == "__main__":
    main()
