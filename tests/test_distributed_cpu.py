"""N>1 path on CPU: world_size-2 `gloo` processes exercise the ray sharding + gather of
`parallel.render_sharded` and the gradient averaging of `parallel.GradSync` (both the flat-buffer
branch the HIP backward produces and the generic branch) — the host logic that `bench.py --gpus N`
and the 8-GPU configs use with backend nccl (= RCCL).  No HIP compute is called here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_pl_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_render(rays):
    # a deterministic per-ray function, so that sharded == unsharded can be checked exactly
    return {"rgb_fine": torch.stack([rays[:, 0] * 2, rays[:, 1] + 1, rays[:, 2] ** 2], 1),
            "depth_fine": rays[:, 6] + rays[:, 7], "opacity_fine": rays.sum(1)}


class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(5, 7)
        self.b = torch.nn.Linear(7, 3)


def _worker(rank, world, port, n_rays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        rays = torch.rand(n_rays, 8, generator=g)
        out = parallel.render_sharded(_fake_render, rays)
        ref = _fake_render(rays)
        ok_render = all(torch.equal(out[k], ref[k]) for k in ref)

        # --- GradSync, generic branch: per-rank grads r+1 -> mean (world+1)/2
        m = _Tiny()
        for p in m.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        parallel.GradSync([m]).sync()
        want = (world + 1) / 2.0
        ok_generic = all(torch.allclose(p.grad, torch.full_like(p, want)) for p in m.parameters())

        # --- GradSync, flat branch: grads are views of one flat buffer (what ops.mlp_bwd hands autograd)
        m2 = _Tiny()
        sizes = [p.numel() for p in m2.parameters()]
        flat = torch.arange(sum(sizes), dtype=torch.float32) * (rank + 1)
        off = 0
        for p, sz in zip(m2.parameters(), sizes):
            p.grad = flat[off:off + sz].view_as(p)
            off += sz
        m2._flat_grad = flat
        parallel.GradSync([m2]).sync()
        ok_flat = torch.allclose(flat, torch.arange(sum(sizes), dtype=torch.float32) * want)
        # and the parameters' .grad still alias the averaged buffer
        ok_alias = all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in m2.parameters())
        q.put((rank, ok_render, ok_generic, ok_flat, ok_alias))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rays", [10, 7, 1])      # even split, ragged split, fewer rays than ranks
def test_world2_gloo_sharding_and_gradsync(n_rays):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 640000, 190512):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    # BASELINE configs[4]: 800x800 image over 8 GPUs = 80,000 rays each
    assert parallel.shard_bounds(640000, 3, 8) == (240000, 320000)


def test_single_process_passthrough():
    rays = torch.rand(5, 8)
    out = parallel.render_sharded(_fake_render, rays, keys=("rgb_fine",))
    assert torch.equal(out["rgb_fine"], _fake_render(rays)["rgb_fine"])
    parallel.GradSync([_Tiny()]).sync()     # no process group: no-op
