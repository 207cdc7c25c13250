"""Multi-GPU plumbing for the hot path: one process per GPU (`torch.distributed`, backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

* Rays are independent units: inference shards the ray list contiguously across ranks with no
  collective on the data path and one gather of the finished pixels (eval.py:58-86 sharded).
* Training is the reference's DDP (train.py:174-175): replicated models, each rank draws its own
  batch, gradients are averaged.  Each model's 24 gradients live in ONE flat fp32 buffer written by
  the dW-reduce kernel (2.38 MB), so a step needs exactly one all-reduce per model — a message that on
  the 8-GPU xGMI mesh is latency-bound (~10 us of wire time) and is issued for the fine model while
  the coarse model's backward is still running (autograd runs fine first).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) of `n` rays owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def render_sharded(render_fn, rays, keys=("rgb_fine", "depth_fine", "opacity_fine"), group=None):
    """Ray-sharded full-image inference: every rank renders its contiguous slice with
    `render_fn(rays_slice) -> dict`, then the requested keys are all-gathered (padded to equal length)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = rays.shape[0]
    lo, hi = shard_bounds(n, rank, world)
    local = render_fn(rays[lo:hi])
    if world == 1:
        return {k: local[k] for k in keys if k in local}
    out = {}
    maxlen = shard_bounds(n, 0, world)[1]
    for k in keys:
        if k not in local:
            continue
        v = local[k]
        pad = torch.zeros((maxlen,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[: v.shape[0]] = v
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        parts = []
        for r in range(world):
            a, b = shard_bounds(n, r, world)
            parts.append(bufs[r][: b - a])
        out[k] = torch.cat(parts, 0)
    return out


class GradSync:
    """Average gradients across ranks: one all-reduce per model on its flat gradient buffer when the
    HIP backward produced one, else a flatten/all-reduce/unflatten of the parameter grads."""

    def __init__(self, models, group=None, force=False):
        self.models = list(models)
        self.group = group
        self.force = force          # run the collectives even at world size 1 (tests the RCCL path on one GPU)

    def sync(self):
        if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and not self.force):
            return
        world = dist.get_world_size(self.group)
        works = []
        for m in self.models:
            flat = getattr(m, "_flat_grad", None)
            params = [p for p in m.parameters() if p.grad is not None]
            if (flat is not None and params and params[0].grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
                    and params[-1].grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()):
                works.append((dist.all_reduce(flat, group=self.group, async_op=True), flat, None))
            elif params:
                buf = torch.cat([p.grad.reshape(-1) for p in params])
                works.append((dist.all_reduce(buf, group=self.group, async_op=True), buf, params))
        for w, buf, params in works:
            w.wait()
            buf.div_(world)
            if params is not None:
                off = 0
                for p in params:
                    n = p.grad.numel()
                    p.grad.copy_(buf[off:off + n].view_as(p.grad))
                    off += n
