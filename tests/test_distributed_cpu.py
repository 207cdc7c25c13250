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


class _RecordingBackend:
    """Stand-in for system._HipGraphBackend on CPU.  Like a stream capture, `capture` does NOT execute the work (it only
    records the callable); every `replay` executes it and refreshes the static output holder in place — the same
    contract (static inputs, static outputs, nothing runs at capture time) without a device."""

    class _G:
        def __init__(self, fn, holder):
            self.fn, self.holder, self.replays = fn, holder, 0

        def replay(self):
            self.replays += 1
            r = self.fn()
            if isinstance(r, dict):
                self.holder.clear()
                self.holder.update(r)

    def __init__(self):
        self.captures = 0

    def on_side_stream(self, fn, *args):
        return fn(*args)

    def capture(self, fn, share_pool_with=None):
        self.captures += 1
        holder = {}
        return self._G(fn, holder), holder


class _TinySystem(torch.nn.Module):
    """training_step contract of NeRFSystem (train.py:103-117) on a toy model whose backward announces its flat gradient
    buffer exactly like the fused HIP backward does (models/mlp_autograd._param_grads)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.net = _Tiny()
        self.models = [self.net]

    def training_step(self, batch, nb):
        y = self.net.b(torch.relu(self.net.a(batch["x"])))
        return {"loss": ((y - batch["y"]) ** 2).mean()}


def _graphed_step_host_logic(rank, world):
    from nerf_pl_amd.system import GraphedTrainStep
    g = torch.Generator().manual_seed(100 + rank)           # every rank draws its own batch (weak scaling)
    batches = [{"x": torch.randn(16, 5, generator=g), "y": torch.randn(16, 3, generator=g)} for _ in range(7)]
    sysm = _TinySystem()
    opt = torch.optim.SGD(sysm.parameters(), lr=0.1)
    sync = parallel.GradSync(sysm.models)
    be = _RecordingBackend()
    stepper = GraphedTrainStep(sysm, opt, grad_sync=sync, warmup=2, backend=be, sync_in_graph=False)    # the two-graph form
    for b in batches[:5]:
        stepper(b)
    ok = be.captures == 2 and stepper.graph is not None and stepper.graph_opt is not None      # two graphs
    ok = ok and stepper.graph.replays == 3 and stepper.graph_opt.replays == 3                    # calls 3,4,5 replayed
    ok = ok and sync.hooks_enabled
    for grp in opt.param_groups:                            # scheduler step: lr is a captured argument
        grp["lr"] = 0.05
    for b in batches[5:]:
        stepper(b)
    ok = ok and be.captures == 4 and stepper.captured_lr == 0.05
    # replicas stay identical (same init, averaged gradients) and equal a single-process run on the averaged gradient
    flat = torch.cat([p.detach().reshape(-1) for p in sysm.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
    ref = _TinySystem()
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1)
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    allb = [[{"x": torch.randn(16, 5, generator=gg), "y": torch.randn(16, 3, generator=gg)} for _ in range(7)] for gg in gens]
    for i in range(7):
        if i == 5:
            for grp in ropt.param_groups:
                grp["lr"] = 0.05
        ropt.zero_grad()
        for r in range(world):
            (ref.training_step(allb[r][i], i)["loss"] / world).backward()
        ropt.step()
    rflat = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    ok = bool(ok and torch.allclose(flat, rflat, rtol=1e-5, atol=1e-6))
    # the default IS the one-graph form (round 6; the two-graph form is the agreed fallback and an explicit choice)
    ok = ok and not GraphedTrainStep(sysm, opt, grad_sync=sync, backend=_RecordingBackend())._two_graphs()
    # the default N > 1 form: ONE graph with the collectives inside (issued by sync() during the captured step) — same replicas
    sys1 = _TinySystem()
    opt1 = torch.optim.SGD(sys1.parameters(), lr=0.1)
    be1 = _RecordingBackend()
    step1 = GraphedTrainStep(sys1, opt1, grad_sync=parallel.GradSync(sys1.models), warmup=2, backend=be1)
    for i, b in enumerate(batches):
        if i == 5:
            for grp in opt1.param_groups:
                grp["lr"] = 0.05
        step1(b)
    flat1 = torch.cat([p.detach().reshape(-1) for p in sys1.parameters()])
    ok = ok and be1.captures == 2 and step1.graph_opt is None and step1.graph.replays == 2        # one graph per capture
    ok = bool(ok and torch.allclose(flat1, rflat, rtol=1e-5, atol=1e-6))
    # a stack that cannot capture the collective: the one-graph capture raises, the stepper falls back to two graphs on every
    # rank alike and the replicas still follow the reference run
    sys2 = _TinySystem()
    opt2 = torch.optim.SGD(sys2.parameters(), lr=0.1)

    class _NoCollectiveInGraph(_RecordingBackend):
        def capture(self, fn, share_pool_with=None):
            if getattr(fn, "__name__", "") == "<lambda>" and "_eager" in fn.__code__.co_names:
                raise RuntimeError("collective not capturable")
            return super().capture(fn, share_pool_with)

    be2 = _NoCollectiveInGraph()
    step2 = GraphedTrainStep(sys2, opt2, grad_sync=parallel.GradSync(sys2.models), warmup=2, backend=be2, sync_in_graph=True)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i, b in enumerate(batches):
            if i == 5:
                for grp in opt2.param_groups:
                    grp["lr"] = 0.05
            step2(b)
    flat2 = torch.cat([p.detach().reshape(-1) for p in sys2.parameters()])
    ok = ok and step2.capture_fallback is not None and step2._two_graphs() and step2.graph_opt is not None
    ok = bool(ok and torch.allclose(flat2, rflat, rtol=1e-5, atol=1e-6))
    # ... and a capture that fails on ONE rank only (rank 1), with its peer waiting in the collective the ranks use to agree on the
    # branch (GradSync.agree_any): BOTH ranks must fall back — a rank replaying a captured all-reduce against a rank issuing an
    # eager one would deadlock — and the replicas still follow the reference run
    sys3 = _TinySystem()
    opt3 = torch.optim.SGD(sys3.parameters(), lr=0.1)

    class _Rank1CannotCapture(_RecordingBackend):
        def capture(self, fn, share_pool_with=None):
            if rank == 1 and getattr(fn, "__name__", "") == "<lambda>" and "_eager" in fn.__code__.co_names:
                raise ValueError("capture failed on this rank only (not a RuntimeError: the peer must still not be left waiting)")
            return super().capture(fn, share_pool_with)

    step3 = GraphedTrainStep(sys3, opt3, grad_sync=parallel.GradSync(sys3.models), warmup=2, backend=_Rank1CannotCapture(), sync_in_graph=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i, b in enumerate(batches):
            if i == 5:
                for grp in opt3.param_groups:
                    grp["lr"] = 0.05
            step3(b)
    flat3 = torch.cat([p.detach().reshape(-1) for p in sys3.parameters()])
    ok = ok and step3.capture_fallback is not None and step3._two_graphs() and step3.graph_opt is not None
    ok = ok and (("another rank" in step3.capture_fallback) == (rank != 1))
    return bool(ok and torch.allclose(flat3, rflat, rtol=1e-5, atol=1e-6))


def _fused_backward_issue_order(rank, world):
    """The fused training node's backward (models/train_step._TrainRender.backward) under the two GradSync forms, with the HIP
    launch (ops.mlp_bwd_multi) replaced by a recorder: WHAT is launched and WHEN the collectives are issued relative to it.
      form="per_model": fine chain/dW/reduce -> the fine model's all-reduce is ISSUED -> only then the coarse model's launches;
      form="merged" (default): ONE launch for both models -> ONE all-reduce over the joint buffer."""
    from types import SimpleNamespace

    from nerf_pl_amd import ops
    from nerf_pl_amd.models import NeRF
    from nerf_pl_amd.models import train_step as TS
    ok = True
    real_multi, real_ar = ops.mlp_bwd_multi, dist.all_reduce
    for form in ("per_model", "merged"):
        torch.manual_seed(3)
        coarse, fine = NeRF(), NeRF()
        gs = parallel.GradSync([coarse, fine], form=form)
        log = []

        def fake_multi(entries, dtype, adam=None, phases=7, workspace=None, g_scale=None):
            log.append(("launch", tuple(e[0] for e in entries)))
            joint = torch.empty(len(entries) * ops.FLAT_GRAD_FLOATS)
            out = []
            for k, e in enumerate(entries):
                fl = joint[k * ops.FLAT_GRAD_FLOATS:(k + 1) * ops.FLAT_GRAD_FLOATS]
                fl.fill_(float(rank + 1) * (1.0 if e[0] == "fine" else 3.0))
                out.append(ops.flat_grad_views(1, "cpu", out=fl))
            return out

        def logging_ar(t, *a_, **kw_):
            log.append(("all_reduce", t.numel()))
            return real_ar(t, *a_, **kw_)
        ops.mlp_bwd_multi, dist.all_reduce = fake_multi, logging_ar
        try:
            ctx = SimpleNamespace(n_params=[24, 24], entries=[("fine", None, None, None), ("coarse", None, None, None)], dtype="bf16",
                                  adam=None, models=[fine, coarse], serials=[0, 0])
            grads = TS._TrainRender.backward(ctx, torch.ones(()))
            # autograd adopts the returned views as p.grad (parameter order of forward(): coarse, then fine)
            for p_, g_ in zip(coarse.flat_params() + fine.flat_params(), grads[3:]):
                p_.grad = g_
            issued = list(log)
            gs.sync()
        finally:
            ops.mlp_bwd_multi, dist.all_reduce = real_multi, real_ar
        n = ops.FLAT_GRAD_FLOATS
        if form == "per_model":
            ok = ok and issued == [("launch", ("fine",)), ("all_reduce", n), ("launch", ("coarse",)), ("all_reduce", n)]
            ok = ok and gs.issue_log[:2] == [("model", id(fine)), ("model", id(coarse))]
        else:
            ok = ok and issued == [("launch", ("fine", "coarse")), ("all_reduce", 2 * n)] and gs.issue_log[:1] == [("joint", 2)]
        ok = ok and len(log) == len(issued)                                   # sync() only waited: no further collective
        want = (world + 1) / 2.0
        ok = ok and all(torch.allclose(p_.grad, torch.full_like(p_, want)) for p_ in fine.parameters())
        ok = ok and all(torch.allclose(p_.grad, torch.full_like(p_, 3.0 * want)) for p_ in coarse.parameters())
        gs.detach()
    return bool(ok)


def _worker(rank, world, port, n_rays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        rays = torch.rand(n_rays, 8, generator=g)
        calls = {"n": 0}
        real_gather = dist.all_gather_into_tensor

        def counting_gather(*a, **kw):
            calls["n"] += 1
            return real_gather(*a, **kw)
        dist.all_gather_into_tensor = counting_gather
        try:
            out = parallel.render_sharded(_fake_render, rays)
        finally:
            dist.all_gather_into_tensor = real_gather
        ref = _fake_render(rays)
        # every key comes back whole, through ONE collective per image (rgb + depth + opacity packed as the columns of one buffer)
        ok_render = all(torch.equal(out[k], ref[k]) for k in ref) and calls["n"] == 1

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
        # --- GradSync, overlap: the fused backward announces a finished flat buffer through the grad-ready hook; the
        # all-reduce is issued THEN (while "the rest of backward" still runs) and sync() only waits and averages
        m3 = _Tiny()
        gs = parallel.GradSync([m3])
        base3 = torch.arange(sum(sizes), dtype=torch.float32)

        def adopt(model, flat_):                            # what autograd does with the views the backward returns
            off_ = 0
            for p_, sz_ in zip(model.parameters(), sizes):
                p_.grad = flat_[off_:off_ + sz_].view_as(p_)
                off_ += sz_
        flat3 = base3 * (rank + 1) + 1.0
        m3._flat_grad = flat3
        assert m3._grad_ready_hook is not None
        m3._grad_ready_hook(m3, flat3)                      # what models/mlp_autograd._param_grads does (p.grad still None)
        adopt(m3, flat3)
        started = gs.started_early
        gs.sync()
        ok_overlap = started == 1 and not gs._inflight and torch.allclose(flat3, base3 * want + 1.0)
        # hooks switched off (two-graph capture): nothing is issued early, sync() does the whole job
        gs.hooks_enabled = False
        for p_ in m3.parameters():
            p_.grad = None
        flat3.copy_(base3 * (rank + 1))
        m3._grad_ready_hook(m3, flat3)
        adopt(m3, flat3)
        ok_overlap = ok_overlap and gs.started_early == 1
        gs.sync()
        ok_overlap = ok_overlap and torch.allclose(flat3, base3 * want)
        gs.hooks_enabled = True
        # gradient accumulation: p.grad exists when the backward announces its buffer -> autograd ACCUMULATES the
        # views into p.grad, so no early collective may touch the buffer; sync() averages the accumulated p.grad
        acc = [torch.full_like(p_, 10.0 * (rank + 1)) for p_ in m3.parameters()]
        for p_, a_ in zip(m3.parameters(), acc):
            p_.grad = a_.clone()
        flat4 = base3 * (rank + 1)
        m3._flat_grad = flat4
        before = gs.started_early
        m3._grad_ready_hook(m3, flat4)
        off = 0
        for p_, sz in zip(m3.parameters(), sizes):
            p_.grad += flat4[off:off + sz].view_as(p_)
            off += sz
        gs.sync()
        got = torch.cat([p_.grad.reshape(-1) for p_ in m3.parameters()])
        ok_overlap = ok_overlap and gs.started_early == before and torch.allclose(got, base3 * want + 10.0 * want)
        # two backwards before one sync(): the first buffer is all-reduced early, the second backward accumulates on top of it
        for p_ in m3.parameters():
            p_.grad = None
        flat5 = base3 * (rank + 1)
        m3._flat_grad = flat5
        m3._grad_ready_hook(m3, flat5)
        adopt(m3, flat5)
        flat6 = base3 * (rank + 1) * 2.0
        m3._flat_grad = flat6
        m3._grad_ready_hook(m3, flat6)                      # waits for + finishes the first collective, issues nothing
        off = 0
        for p_, sz in zip(m3.parameters(), sizes):
            p_.grad += flat6[off:off + sz].view_as(p_)
            off += sz
        gs.sync()
        got = torch.cat([p_.grad.reshape(-1) for p_ in m3.parameters()])
        ok_overlap = ok_overlap and torch.allclose(got, base3 * want * 3.0)
        # autograd COPIED the announced views instead of adopting them: refused, never silently un-averaged
        for p_ in m3.parameters():
            p_.grad = None
        flat7 = base3 * (rank + 1)
        m3._flat_grad = flat7
        m3._grad_ready_hook(m3, flat7)
        adopt(m3, flat7.clone())
        try:
            gs.sync()
            ok_overlap = False
        except RuntimeError:
            pass

        # --- round 5: the two-graph step's sync() issues the models' all-reduces as ONE grouped launch (RCCL: ncclGroupStart / End
        # through torch's coalescing manager).  Same branch driven over gloo here: one manager entry for both buffers, the values
        # of two separate calls; and when the grouped form raises, sync() falls back to one call per model for good.
        def two_models():
            ms_, flats_ = [], []
            for k in range(2):
                mm = _Tiny()
                fl = base3 * (rank + 1) * (k + 1)
                adopt(mm, fl)
                mm._flat_grad = fl
                ms_.append(mm)
                flats_.append(fl)
            return ms_, flats_
        ms8, flats8 = two_models()
        gs8 = parallel.GradSync(ms8)
        gs8.hooks_enabled = False
        gs8._coalesce_backends = ("nccl", "gloo")
        entered = {"n": 0}
        real_cm = dist._coalescing_manager

        def counting_cm(*a, **kw):
            entered["n"] += 1
            return real_cm(*a, **kw)
        dist._coalescing_manager = counting_cm
        try:
            gs8.sync()
        finally:
            dist._coalescing_manager = real_cm
        ok_grouped = entered["n"] == 1 and gs8._coalesce and all(torch.allclose(f, base3 * want * (k + 1)) for k, f in enumerate(flats8))

        def broken_cm(*a, **kw):                                # a stack WITHOUT the grouped form: the only reason to fall back
            raise NotImplementedError("no grouped collectives on this stack")
        ms9, flats9 = two_models()
        gs9 = parallel.GradSync(ms9)
        gs9.hooks_enabled = False
        gs9._coalesce_backends = ("nccl", "gloo")
        dist._coalescing_manager = broken_cm
        try:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                gs9.sync()
        finally:
            dist._coalescing_manager = real_cm
        ok_grouped = ok_grouped and not gs9._coalesce and all(torch.allclose(f, base3 * want * (k + 1)) for k, f in enumerate(flats9))
        #  any OTHER error — a communicator error, a failure after a partial enqueue — is re-raised, never papered over by
        # re-issuing the collectives one by one (they could run twice, or mismatch across the ranks): every rank raises alike here
        def failing_cm(*a, **kw):
            raise RuntimeError("communicator error")
        ms10, flats10 = two_models()
        gs10 = parallel.GradSync(ms10)
        gs10.hooks_enabled = False
        gs10._coalesce_backends = ("nccl", "gloo")
        dist._coalescing_manager = failing_cm
        try:
            gs10.sync()
            ok_grouped = False
        except RuntimeError:
            ok_grouped = ok_grouped and gs10._coalesce and all(torch.allclose(f, base3 * (rank + 1) * (k + 1)) for k, f in enumerate(flats10))
        finally:
            dist._coalescing_manager = real_cm
        ok_overlap = ok_overlap and ok_grouped

        # --- round 6: the fused step's backward writes both models' flat buffers as consecutive slices of ONE allocation
        # (ops.mlp_bwd_multi) and announces them together; GradSync(form="merged") — the default — issues ONE all-reduce over the
        # joint buffer from that hook, and sync() only waits.  With the hooks off (the two-graph fallback) sync() itself sends the
        # joint range as one message: no coalescing manager involved.
        def joint_models(scale):
            ms_ = [_Tiny(), _Tiny()]
            n_ = sum(sizes)
            joint_ = torch.empty(2 * n_)
            flats_ = [joint_[k * n_:(k + 1) * n_] for k in range(2)]
            for k, fl in enumerate(flats_):
                fl.copy_(base3 * (rank + 1) * (k + 1) * scale)
                ms_[k]._flat_grad = fl
            return ms_, flats_, joint_
        ms11, flats11, joint11 = joint_models(1.0)
        gs11 = parallel.GradSync(ms11)
        calls11 = {"n": 0, "numel": []}
        real_ar = dist.all_reduce

        def counting_ar(t, *a_, **kw_):
            calls11["n"] += 1
            calls11["numel"].append(t.numel())
            return real_ar(t, *a_, **kw_)
        dist.all_reduce = counting_ar
        try:
            ok_joint = gs11.form == "merged" and ms11[0]._grads_ready_hook is not None and ms11[0]._grads_ready_hook == ms11[1]._grads_ready_hook
            ms11[0]._grads_ready_hook(ms11, flats11)             # what models/train_step._TrainRender.backward does (p.grad still None)
            ok_joint = ok_joint and calls11["n"] == 1 and calls11["numel"] == [2 * sum(sizes)] and gs11.issue_log == [("joint", 2)]
            for mm, fl in zip(ms11, flats11):
                adopt(mm, fl)
            gs11.sync()
            ok_joint = ok_joint and calls11["n"] == 1 and not gs11._inflight       # sync() only waited
            ok_joint = ok_joint and all(torch.allclose(f, base3 * want * (k + 1)) for k, f in enumerate(flats11))
            # hooks off: sync() sends the joint range itself, as ONE message
            ms12, flats12, _ = joint_models(2.0)
            gs12 = parallel.GradSync(ms12)
            gs12.hooks_enabled = False
            gs12._coalesce_backends = ("nccl", "gloo")
            entered["n"] = 0
            dist._coalescing_manager = counting_cm
            calls11["n"], calls11["numel"] = 0, []
            ms12[0]._grads_ready_hook(ms12, flats12)
            for mm, fl in zip(ms12, flats12):
                adopt(mm, fl)
            gs12.sync()
            ok_joint = ok_joint and calls11["n"] == 1 and calls11["numel"] == [2 * sum(sizes)] and entered["n"] == 0
            ok_joint = ok_joint and all(torch.allclose(f, base3 * want * (k + 1) * 2.0) for k, f in enumerate(flats12))
            # gradient accumulation under the joint hook: nothing is issued early, sync() averages the accumulated p.grad
            ms13, flats13, _ = joint_models(1.0)
            gs13 = parallel.GradSync(ms13)
            for mm in ms13:
                for p_ in mm.parameters():
                    p_.grad = torch.full_like(p_, 10.0 * (rank + 1))
            calls11["n"] = 0
            ms13[0]._grads_ready_hook(ms13, flats13)
            ok_joint = ok_joint and calls11["n"] == 0
            for mm, fl in zip(ms13, flats13):
                off = 0
                for p_, sz in zip(mm.parameters(), sizes):
                    p_.grad += fl[off:off + sz].view_as(p_)
                    off += sz
            gs13.sync()
            got13 = [torch.cat([p_.grad.reshape(-1) for p_ in mm.parameters()]) for mm in ms13]
            ok_joint = ok_joint and all(torch.allclose(g_, base3 * want * (k + 1) + 10.0 * want) for k, g_ in enumerate(got13))
        finally:
            dist.all_reduce = real_ar
            dist._coalescing_manager = real_cm
        ok_overlap = ok_overlap and ok_joint
        ok_overlap = ok_overlap and _fused_backward_issue_order(rank, world)

        # --- the N>1 training step's host logic (system.GraphedTrainStep): eager warm-up steps, capture of
        # [forward+backward] and [optimizer] as two graphs with the collective issued eagerly in between, replays,
        # re-capture on a learning-rate change — with a recording stand-in for the hipGraph backend.
        ok_step = _graphed_step_host_logic(rank, world)
        q.put((rank, ok_render, ok_generic, ok_flat, ok_alias, ok_overlap, ok_step))
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


def test_graphed_step_batch_source_host_logic():
    """GraphedTrainStep(batch_source=...): the source is called inside every eager step and inside every replay (i.e. inside
    the captured region), never at capture time, and mixing it with an explicit batch is refused."""
    from nerf_pl_amd.system import GraphedTrainStep
    sysm = _TinySystem()
    opt = torch.optim.SGD(sysm.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(3)
    calls = []

    def source():
        calls.append(len(calls))
        return {"x": torch.randn(16, 5, generator=g), "y": torch.randn(16, 3, generator=g)}
    be = _RecordingBackend()
    stepper = GraphedTrainStep(sysm, opt, warmup=2, backend=be, batch_source=source)
    before = torch.cat([p.detach().reshape(-1).clone() for p in sysm.parameters()])
    for _ in range(5):
        stepper()
    assert be.captures == 1 and stepper.graph.replays == 3
    assert len(calls) == 5                                  # 2 eager + 3 replays; the capture itself ran nothing
    after = torch.cat([p.detach().reshape(-1) for p in sysm.parameters()])
    assert not torch.equal(before, after)
    with pytest.raises(ValueError):
        stepper({"x": torch.zeros(16, 5), "y": torch.zeros(16, 3)})
    with pytest.raises(ValueError):
        GraphedTrainStep(sysm, opt, warmup=2, backend=_RecordingBackend())()


def _torchrun(nproc, script_args, timeout=180):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "tests", "host", "bench_rank_logic.py")] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=root)


def test_bench_rank_logic_world2_gloo_end_to_end():
    """bench.py's own launch-contract functions under the launcher the driver uses (torch.distributed.run, 2 processes, gloo):
    init_world sees a 2-rank communicator, timed_region reports the SLOWER rank's time (rank 1 sleeps twice as long), the
    sharded image equals the unsharded one, and exactly one JSON line comes out (rank 0's)."""
    import json
    r = _torchrun(2, ["2"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_nranks"] == 2 and d["sharded_equal"] and d["span"] == [0, 501]
    assert 5 * 0.04 * 0.95 <= d["dt"] <= 5 * 0.04 * 3, d["dt"]          # 5 steps of rank 1's 40 ms, not rank 0's 20 ms


def test_bench_rank_logic_refuses_a_world_that_is_not_gpus():
    """2 processes launched, --gpus 4 requested: every rank exits 2 before any group is formed, no line."""
    r = _torchrun(2, ["4"])
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "WORLD_SIZE=2 but --gpus 4" in r.stderr
