"""Multi-GPU plumbing for the hot path: one process per GPU (`torch.distributed`, backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

* Rays are independent units: inference shards the ray list contiguously across ranks with no
  collective on the data path and one gather of the finished pixels (eval.py:58-86 sharded).
* Training is the reference's DDP (train.py:174-175): replicated models, each rank draws its own
  batch, gradients are averaged.  Each model's 24 gradients live in ONE flat fp32 buffer written by
  the dW-reduce kernel (2.38 MB), and the fused step's backward writes both models' buffers as consecutive
  slices of one allocation: a step needs exactly ONE all-reduce (4.77 MB) — a message that on the 8-GPU xGMI
  mesh is latency-bound.  `GradSync(form="merged")` (the default) keeps the one-rank step's launches (one chain, one dW,
  one reduce launch for both models) and issues that all-reduce from the backward's grads-ready hook;
  `form="per_model"` runs the two models' backwards one after the other and issues the fine model's all-reduce while the
  coarse model's backward is still running (autograd runs the fine model first; `GradSync._on_grad_ready`) — what DDP's
  bucket hooks give the reference, at the price of un-merging the launches (7.6 % of the step at world 1).
"""
import os

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
    # ONE collective per image: the requested keys travel as the columns of one (maxlen, C) buffer (rgb 3 + depth 1 + opacity 1
    # = 5 floats per ray; at 8 ranks the 0.96 MB gather is latency-bound, so the count of collectives is what matters)
    have = [k for k in keys if k in local]
    if not have:
        return {}
    cols = [int(torch.Size(local[k].shape[1:]).numel()) for k in have]          # (rows, ...) -> columns per row (1 for a vector)
    maxlen = shard_bounds(n, 0, world)[1]
    v0 = local[have[0]]
    pack = torch.zeros(maxlen, sum(cols), dtype=v0.dtype, device=v0.device)
    c = 0
    for k, w in zip(have, cols):
        pack[: hi - lo, c:c + w] = local[k].reshape(hi - lo, w)
        c += w
    gathered = torch.empty(world * maxlen, sum(cols), dtype=v0.dtype, device=v0.device)
    dist.all_gather_into_tensor(gathered, pack, group=group)
    rows = torch.cat([gathered[r * maxlen: r * maxlen + (shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0])]
                      for r in range(world)], 0)
    out, c = {}, 0
    for k, w in zip(have, cols):
        col = rows[:, c:c + w]
        out[k] = col.reshape((n,) + tuple(local[k].shape[1:])) if local[k].dim() > 1 else col.reshape(n)
        c += w
    return out


_COALESCE = os.environ.get("NERFHIP_COALESCE_ALLREDUCE", "1") != "0"      # (A/B switch)


class _Done:
    """A finished piece of work (its collective was waited for as part of a group)."""

    def wait(self):
        return True


class GradSync:
    """Average gradients across ranks (the reference's DDP, train.py:174-175): one all-reduce per model on the flat
    gradient buffer the dW-reduce kernel wrote (2.38 MB), else a flatten/all-reduce/unflatten of the parameter grads.

    Overlap with backward (what DDP's bucket hooks give the reference): `attach()` registers a grad-ready hook on each
    model; the fused MLP backward calls it the moment the model's flat gradient buffer is complete
    (models/mlp_autograd.py:_param_grads), and the hook issues that model's all-reduce asynchronously on the
    communicator's stream.  Autograd runs the fine model's backward first, so its all-reduce travels over xGMI while
    compositing / importance sampling / the coarse model's backward still execute; `sync()` then waits for whatever
    was started, launches whatever was not, and divides by the world size."""

    def __init__(self, models, group=None, force=False, overlap=True, form=None):
        self.models = list(models)
        self.group = group
        self.force = force          # run the collectives even at world size 1 (tests the RCCL path on one GPU)
        self.overlap = overlap
        # form of the fused step's backward under this sync: "merged" = the one-rank launches + ONE all-reduce of the step's joint
        # gradient buffer; "per_model" = per model chain -> dW -> reduce -> all-reduce (the fine model's travels under the coarse
        # model's backward).  NERFHIP_GRAD_SYNC_FORM overrides the default for A/B runs (tools/launch_scale.sh measures both).
        self.form = form if form is not None else os.environ.get("NERFHIP_GRAD_SYNC_FORM", "merged")
        if self.form not in ("merged", "per_model"):
            raise ValueError("GradSync form must be 'merged' or 'per_model', got %r" % (self.form,))
        self.issue_log = []         # ("model", id(model)) / ("joint", n_models) per hook-issued collective since the last sync() (tests)
        self.hooks_enabled = True   # GraphedTrainStep switches the hooks off while it captures/replays a graph that
                                    # must not contain the collective (two-graph mode)
        self._inflight = {}         # id(model) -> [(work, flat, needs_division)] issued from the hook since the last sync()
        self.started_early = 0      # statistics: all-reduces issued from the hook (tests assert on it)
        self._coalesce = _COALESCE  # the per-model all-reduces of sync() as one grouped RCCL launch
        self._coalesce_backends = ("nccl",)     # (the gloo CPU test adds "gloo" to drive the same branch)
        if overlap:
            self.attach()

    # -- plumbing ---------------------------------------------------------------------------------------------------
    def active(self):
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    def attach(self):
        for m in self.models:
            m._grad_ready_hook = self._on_grad_ready
            m._grads_ready_hook = self._on_grads_ready if self.form == "merged" else None

    def detach(self):
        for m in self.models:
            if getattr(m, "_grad_ready_hook", None) == self._on_grad_ready:
                m._grad_ready_hook = None
            if getattr(m, "_grads_ready_hook", None) == self._on_grads_ready:
                m._grads_ready_hook = None

    def agree_any(self, flag):
        """True on every rank if `flag` is true on any (one small eager MAX all-reduce): ranks that must take the same branch —
        e.g. GraphedTrainStep falling back to two graphs when the one-graph capture failed somewhere — decide it here."""
        if not self.active():
            return bool(flag)
        dev = next(self.models[0].parameters()).device
        t = torch.tensor([1 if flag else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def _avg_op(self):
        """(reduce op, needs_division): RCCL averages in the collective itself; gloo (CPU tests) has no AVG."""
        backend = dist.get_backend(self.group)
        if backend == "nccl" and hasattr(dist.ReduceOp, "AVG"):
            return dist.ReduceOp.AVG, False
        return dist.ReduceOp.SUM, True

    def _on_grad_ready(self, model, flat):
        """Called by the fused backward when `flat` (the model's 24 gradients) is complete, BEFORE autograd hands the views
        to the parameters.  The all-reduce is started here only when autograd is going to adopt the views as `p.grad`
        (every `p.grad` is None: zero_grad(set_to_none=True), no accumulation): otherwise autograd would accumulate the views
        into existing `.grad` tensors on the compute stream while the collective rewrites `flat` on the communicator's."""
        if not (self.hooks_enabled and self.overlap and self.active()):
            return
        # a second backward before sync(): its accumulation into p.grad (views of the first buffer) must be ordered after
        # the collective that is still rewriting that buffer — finish (wait, divide) what is in flight for this model
        self._finish(self._inflight.get(id(model), ()))
        if any(p.grad is not None for p in model.parameters()):
            return                                     # gradient accumulation: sync() reduces the accumulated p.grad
        op, div = self._avg_op()
        work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        self._inflight.setdefault(id(model), []).append([work, flat, div, False])
        self.started_early += 1
        self.issue_log.append(("model", id(model)))

    @staticmethod
    def joint_of(flats):
        """The ONE tensor whose consecutive slices `flats` are (ops.mlp_bwd_multi allocates a step's flat gradient buffers that
        way), or None."""
        base = getattr(flats[0], "_base", None)
        if base is None or any(getattr(f, "_base", None) is not base for f in flats) or not base.is_contiguous():
            return None
        off = flats[0].storage_offset()
        for f in flats:
            if f.storage_offset() != off or not f.is_contiguous():
                return None
            off += f.numel()
        lo = flats[0].storage_offset() - base.storage_offset()
        return base.reshape(-1)[lo:lo + sum(f.numel() for f in flats)]

    def _on_grads_ready(self, models, flats):
        """form="merged": called ONCE by the fused backward when the flat gradient buffers of ALL its models are complete (one
        reduce launch wrote them), BEFORE autograd hands the views to the parameters.  Issues ONE all-reduce over the joint buffer,
        asynchronously on the communicator's stream; the same adoption rule as `_on_grad_ready`."""
        if not (self.hooks_enabled and self.overlap and self.active()):
            return
        for m in models:
            self._finish(self._inflight.get(id(m), ()))
        self._finish(self._inflight.get("joint", ()))
        if any(p.grad is not None for m in models for p in m.parameters()):
            return                                     # gradient accumulation: sync() reduces the accumulated p.grad
        joint = self.joint_of(flats)
        if joint is None:                              # not one allocation (an older caller): one collective per model
            for m, f in zip(models, flats):
                self._on_grad_ready(m, f)
            return
        op, div = self._avg_op()
        work = dist.all_reduce(joint, op=op, group=self.group, async_op=True)
        self._inflight.setdefault("joint", []).append([work, joint, div, False, [id(m) for m in models], list(flats)])
        self.started_early += 1
        self.issue_log.append(("joint", len(models)))

    def _finish(self, entries):
        """wait for hook-issued collectives and apply their division: afterwards each buffer holds the rank average"""
        for e in entries:
            if not e[3]:
                e[0].wait()
                if e[2]:
                    e[1].div_(dist.get_world_size(self.group))
                e[3] = True

    @staticmethod
    def _aliases(params, flat):
        """autograd adopted the views of `flat` as the parameters' .grad (first and last parameter checked)"""
        sp = flat.untyped_storage().data_ptr()
        return (params[0].grad.untyped_storage().data_ptr() == sp and params[-1].grad.untyped_storage().data_ptr() == sp)

    # -- the step-level call ----------------------------------------------------------------------------------------
    def sync(self):
        if not self.active():
            self._inflight.clear()
            return
        world = dist.get_world_size(self.group)
        op, div = self._avg_op()
        works, direct = [], []
        del self.issue_log[:-64]                        # (a diagnostic: keep the tail)
        covered = set()
        early_joint = self._inflight.pop("joint", [])
        if len(early_joint) == 1 and not early_joint[0][3]:
            work, joint, jdiv, _, mids, flats = early_joint[0]
            by_id = {id(m): m for m in self.models}
            ok = all(i in by_id for i in mids)
            for i, f in zip(mids, flats):
                ps = [p for p in by_id[i].parameters() if p.grad is not None] if ok else []
                ok = ok and bool(ps) and getattr(by_id[i], "_flat_grad", None) is f and self._aliases(ps, f) and not self._inflight.get(i)
            if ok:
                works.append((work, joint, None, jdiv))                     # the hook-issued collective IS the step's gradient
                covered.update(mids)
        if not covered:
            # a joint collective whose buffers are not simply p.grad any more (a later backward accumulated on top): finish it,
            # then reduce p.grad below — exact, as for the per-model case
            self._finish(early_joint)
            for e in early_joint:
                for i, f in zip(e[4], e[5]):
                    ps = [p for m in self.models if id(m) == i for p in m.parameters() if p.grad is not None]
                    if ps and not self._aliases(ps, f):
                        raise RuntimeError("GradSync: the gradient buffer all-reduced from the grads-ready hook was not adopted as "
                                           "p.grad (autograd copied it while the collective was in flight); construct "
                                           "GradSync(overlap=False) for this training loop")
        for m in self.models:
            if id(m) in covered:
                continue
            flat = getattr(m, "_flat_grad", None)
            early = self._inflight.pop(id(m), [])
            params = [p for p in m.parameters() if p.grad is not None]
            if (len(early) == 1 and not early[0][3] and flat is not None and early[0][1] is flat and params
                    and self._aliases(params, flat)):
                works.append((early[0][0], flat, None, early[0][2]))        # the overlapped collective IS the gradient
                continue
            # early collectives whose buffer is not simply p.grad: finish them, then reduce p.grad.  If p.grad still views
            # that buffer (a later backward accumulated on top of the averaged values) averaging again is exact — the
            # averaged part is identical on every rank; if autograd COPIED the views instead of adopting them, the copy
            # raced with the collective and p.grad is undefined: refuse.
            self._finish(early)
            for _, eflat, _, _ in early:
                if params and not self._aliases(params, eflat):
                    raise RuntimeError("GradSync: the gradient buffer all-reduced from the grad-ready hook was not adopted as "
                                       "p.grad (autograd copied it while the collective was in flight); construct "
                                       "GradSync(overlap=False) for this training loop")
            if flat is not None and params and self._aliases(params, flat) and not early:
                direct.append(flat)                                      # issued below, together
            elif params:
                buf = torch.cat([p.grad.reshape(-1) for p in params])
                works.append((dist.all_reduce(buf, op=op, group=self.group, async_op=True), buf, params, div))
        # The flat buffers that are reduced as they are (the two-graph step: nothing was started from the hooks).  The fused step's
        # backward wrote them as consecutive slices of one allocation: ONE all-reduce over that range.  Buffers that are not adjacent
        # (a modular-graph step) go over RCCL as ONE grouped launch (ncclGroupStart / End around the per-model all-reduces: same
        # values — each buffer is still its own all-reduce), elsewhere (gloo: the CPU tests) one call each.
        grouped = None
        joint = self.joint_of(direct) if len(direct) > 1 else None
        if joint is not None:
            works.append((dist.all_reduce(joint, op=op, group=self.group, async_op=True), joint, None, div))
            direct = []
        if (len(direct) > 1 and self._coalesce and dist.get_backend(self.group) in self._coalesce_backends
                and hasattr(dist, "_coalescing_manager")):
            # Only a stack WITHOUT the grouped form is a reason to fall back, and only before anything can have launched: the manager
            # launches on __exit__, so an error from inside the block or from the exit (a communicator error, a partial enqueue) is
            # re-raised — re-issuing after it would double or mismatch collectives across the ranks.
            try:
                cm = dist._coalescing_manager(group=self.group, async_ops=True)
                grouped = cm.__enter__()
            except (AttributeError, TypeError, NotImplementedError) as e:
                import warnings
                warnings.warn("GradSync: grouped all-reduce unavailable on this stack (%r); one call per model from now on" % (e,))
                self._coalesce = False
                cm = grouped = None
            if cm is not None:
                try:
                    for flat in direct:
                        dist.all_reduce(flat, op=op, group=self.group)
                except BaseException as e:
                    cm.__exit__(type(e), e, e.__traceback__)
                    raise
                cm.__exit__(None, None, None)
        if grouped is not None:
            works.append((grouped, direct[0], None, False))
            if div:
                for flat in direct:
                    works.append((_Done(), flat, None, True))
        else:
            for flat in direct:
                works.append((dist.all_reduce(flat, op=op, group=self.group, async_op=True), flat, None, div))
        self._inflight.clear()
        for w, buf, params, need_div in works:
            w.wait()
            if need_div:
                buf.div_(world)
            if params is not None:
                off = 0
                for p in params:
                    n = p.grad.numel()
                    p.grad.copy_(buf[off:off + n].view_as(p.grad))
                    off += n
