"""Run under `python -m torch.distributed.run --nproc-per-node 2` by tests/test_distributed_cpu.py (CPU, gloo): drives bench.py's
OWN rank logic — init_world (environment -> process group -> world-size check -> the communicator's own rank count),
timed_region (barrier-bracketed, MAX over ranks) and the ray-sharded image path (parallel.render_sharded) — end to end with two
processes, and prints one JSON line from rank 0 only, like bench.py.  No HIP compute: the "step" sleeps, the "renderer" is a
closed-form function of the rays."""
import json
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nerf_pl_amd import parallel  # noqa: E402


def main():
    gpus = int(sys.argv[1])
    a = Namespace(gpus=gpus, force_dist=False)
    bench.self_launch(a)                                     # refuses (exit 2) when WORLD_SIZE != --gpus
    dist, world, rank, local, nranks = bench.init_world(a, backend="gloo")
    assert dist is not None and world == gpus and nranks == gpus and rank == int(os.environ["RANK"])
    # rank r's step takes (1 + r) * 20 ms: the job's time is the SLOWEST rank's
    dt = bench.timed_region(lambda: time.sleep(0.02 * (1 + rank)), 1, 5, dist, "cpu")
    # configs[4]: the ray list sharded contiguously, every rank renders its span, the pixels are gathered
    n = 1001
    rays = torch.arange(n * 8, dtype=torch.float32).reshape(n, 8) / 100.0

    def render(r):
        return {"rgb_fine": torch.stack([r[:, 0] * 2, r[:, 1] + 1, r[:, 2] ** 2], 1), "depth_fine": r[:, 6] + r[:, 7]}
    got = parallel.render_sharded(render, rays, keys=("rgb_fine", "depth_fine"))
    want = render(rays)
    same = all(torch.equal(got[k], want[k]) for k in want)
    lo, hi = parallel.shard_bounds(n, rank, world)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "rccl_nranks": nranks, "dt": dt, "sharded_equal": bool(same), "span": [lo, hi]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
