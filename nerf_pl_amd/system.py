"""`NeRFSystem` — the reference's LightningModule (train.py:27-148) over the MI355X-native renderer.

Same attribute names, `forward` chunk loop, `training_step` / `validation_step` /
`validation_epoch_end` return contracts and optimizer recipe, so the reference's `train.py` logic is
unchanged; it subclasses pytorch_lightning.LightningModule when that package exists and plain
nn.Module otherwise (Lightning is not installed in this image; `fit()` below is a 30-line stand-in
for Trainer.fit used by tests and bench — harness, not product).
Datasets are injected (`train_dataset` / `val_dataset`): dataset I/O is out of scope.
"""
import os
from collections import defaultdict

import torch
from torch import nn

from .losses import loss_dict
from .metrics import psnr
from .models.nerf import Embedding, NeRF
from .models.rendering import render_rays

try:  # pragma: no cover - not installed here
    from pytorch_lightning import LightningModule as _Base
except Exception:  # noqa: BLE001
    _Base = nn.Module


def get_learning_rate(optimizer):
    for g in optimizer.param_groups:
        return g['lr']


class NeRFSystem(_Base):
    def __init__(self, hparams, train_dataset=None, val_dataset=None):
        super(NeRFSystem, self).__init__()
        self.hparams_ = hparams
        try:
            self.hparams = hparams
        except Exception:  # newer Lightning makes hparams read-only
            pass
        self.loss = loss_dict[getattr(hparams, 'loss_type', 'mse')]()
        self.embedding_xyz = Embedding(3, 10)
        self.embedding_dir = Embedding(3, 4)
        self.embeddings = [self.embedding_xyz, self.embedding_dir]
        self.nerf_coarse = NeRF()
        self.models = [self.nerf_coarse]
        if hparams.N_importance > 0:
            self.nerf_fine = NeRF()
            self.models += [self.nerf_fine]
        self.train_dataset = train_dataset
        self.val_dataset = val_dataset
        self.white_back = getattr(train_dataset, 'white_back', getattr(hparams, 'white_back', False))
        # MI355X specifics (not in the reference): training_step runs render + loss as one autograd node
        # (models/train_step.py) when the recipe allows it; `fuse_adam` additionally applies the FlatAdam update inside the
        # backward's reduce kernel — only for single-GPU steps (no gradient all-reduce between backward and update)
        self.fused_train_step = getattr(hparams, 'fused_train_step', True)
        self.fuse_adam = False

    @property
    def hp(self):
        return self.hparams_

    def decode_batch(self, batch):
        return batch['rays'], batch['rgbs']

    def forward(self, rays):
        """Batched inference on rays in chunks (train.py:49-71)."""
        B = rays.shape[0]
        hp = self.hp
        results = defaultdict(list)
        for i in range(0, B, hp.chunk):
            rendered = render_rays(self.models, self.embeddings, rays[i:i + hp.chunk], hp.N_samples, hp.use_disp,
                                   hp.perturb, hp.noise_std, hp.N_importance, hp.chunk, self.white_back)
            for k, v in rendered.items():
                results[k] += [v]
        for k, v in results.items():
            results[k] = torch.cat(v, 0) if len(v) > 1 else v[0]
        return results

    def configure_optimizers(self):
        """get_optimizer + get_scheduler of the reference (utils/__init__.py:10-53), dispatched on the same hparams
        (`optimizer`, `momentum`, `weight_decay`, `lr_scheduler`, `num_epochs`, `poly_exp`, `decay_step`, `decay_gamma`).
        Recipes this package does not implement raise instead of silently training with a different one."""
        hp = self.hp
        eps = 1e-8
        params = [p for m in self.models for p in m.parameters()]
        name = getattr(hp, 'optimizer', 'adam')
        wd = getattr(hp, 'weight_decay', 0)
        if name == 'adam':
            if params[0].is_cuda and getattr(hp, 'flat_optimizer', True):
                from .optim import FlatAdam       # same Adam math on one flat tensor per model (HIP kernel, 2 launches)
                self.optimizer = FlatAdam(self.models, lr=hp.lr, eps=eps, weight_decay=wd)
            else:
                self.optimizer = torch.optim.Adam(params, lr=hp.lr, eps=eps, weight_decay=wd, fused=params[0].is_cuda)
        elif name == 'sgd':
            self.optimizer = torch.optim.SGD(params, lr=hp.lr, momentum=getattr(hp, 'momentum', 0.9), weight_decay=wd)
        elif name in ('radam', 'ranger'):
            raise NotImplementedError("optimizer %r (utils/optimizers.py of the reference) is outside the hot path this "
                                      "package implements; use 'adam' or 'sgd'" % name)
        else:
            raise ValueError('optimizer not recognized!')

        sched = getattr(hp, 'lr_scheduler', 'steplr')
        if sched == 'steplr':
            scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=list(getattr(hp, 'decay_step', [20])),
                                                             gamma=getattr(hp, 'decay_gamma', 0.1))
        elif sched == 'cosine':
            scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=hp.num_epochs, eta_min=eps)
        elif sched == 'poly':
            n_ep, pexp = hp.num_epochs, getattr(hp, 'poly_exp', 0.9)
            scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lambda epoch: (1 - epoch / n_ep) ** pexp)
        else:
            raise ValueError('scheduler not recognized!')
        if getattr(hp, 'warmup_epochs', 0) > 0:
            raise NotImplementedError("warmup_epochs > 0 (utils/warmup_scheduler.py of the reference) is not implemented")
        return [self.optimizer], [scheduler]

    def training_step(self, batch, batch_nb):
        """train.py:103-117."""
        log = {'lr': get_learning_rate(self.optimizer)}
        rays, rgbs = self.decode_batch(batch)
        if self._fused_step_ok(rays):
            from .models.train_step import render_rays_train
            hp = self.hp
            adam = self.optimizer if (self.fuse_adam and type(self.optimizer).__name__ == 'FlatAdam'
                                      and self._fused_adam_ok()) else None
            # a batch from RayStore.sample(step_draws=...) brings the step's random draws along (made in the launch that drew its
            # pixels); otherwise the fused node makes them itself — either way from torch's generator stream
            draws = batch.get('draws') if isinstance(batch, dict) else None
            packed = batch.get('packed') if isinstance(batch, dict) else None
            results, loss, out3 = render_rays_train(self.models, self.embeddings, rays, rgbs, hp.N_samples, hp.use_disp, hp.perturb,
                                                    hp.noise_std, hp.N_importance, self.white_back, adam=adam, draws=draws, packed=packed)
            self.loss.last = out3
            log['train/loss'] = loss
            psnr_ = out3[1]
        else:
            results = self(rays)
            log['train/loss'] = loss = self.loss(results, rgbs)
            if getattr(self.loss, 'last', None) is not None:      # fused loss kernel already produced the PSNR
                psnr_ = self.loss.last[1]
            else:
                typ = 'fine' if 'rgb_fine' in results else 'coarse'
                with torch.no_grad():
                    psnr_ = psnr(results[f'rgb_{typ}'], rgbs)
        log['train/psnr'] = psnr_
        return {'loss': loss, 'progress_bar': {'train_psnr': psnr_}, 'log': log}

    def _fused_adam_ok(self):
        """Adam inside the backward's reduce kernel is only the reference's step when nothing sits between the gradients and the
        update: one rank (no all-reduce: with several ranks the update would use LOCAL gradients and the replicas diverge).  The
        other condition — no accumulated gradients, or the update would be applied once per backward — is checked where it can be
        known: in the backward itself (models/train_step.py)."""
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def _fused_step_ok(self, rays):
        if not (self.fused_train_step and torch.is_grad_enabled() and rays.is_cuda and rays.dim() == 2
                and 0 < rays.shape[0] <= self.hp.chunk):
            return False
        from .models.train_step import fusable
        return fusable(self.models, self.embeddings, self.loss)

    def validation_step(self, batch, batch_nb):
        """train.py:119-138 (image logging omitted: harness concern)."""
        rays, rgbs = self.decode_batch(batch)
        rays, rgbs = rays.squeeze(), rgbs.squeeze()
        results = self(rays)
        log = {'val_loss': self.loss(results, rgbs)}
        typ = 'fine' if 'rgb_fine' in results else 'coarse'
        log['val_psnr'] = psnr(results[f'rgb_{typ}'], rgbs)
        return log

    def validation_epoch_end(self, outputs):
        mean_loss = torch.stack([x['val_loss'] for x in outputs]).mean()
        mean_psnr = torch.stack([x['val_psnr'] for x in outputs]).mean()
        return {'progress_bar': {'val_loss': mean_loss, 'val_psnr': mean_psnr},
                'log': {'val/loss': mean_loss, 'val/psnr': mean_psnr}}


def fit(system, batches, grad_sync=None, steps=None):
    """Minimal stand-in for Trainer.fit: configure_optimizers -> per batch training_step, backward,
    (gradient all-reduce), optimizer.step.  Returns the list of training_step outputs' losses."""
    (opt,), _ = system.configure_optimizers()
    losses = []
    for i, batch in enumerate(batches):
        if steps is not None and i >= steps:
            break
        out = system.training_step(batch, i)
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()
        if grad_sync is not None:
            grad_sync.sync()
        opt.step()
        losses.append(out['loss'].detach())
    return losses


class GraphedTrainStep:
    """One full training step (training_step -> backward -> [gradient all-reduce] -> optimizer.step) replayed as
    hipGraphs (`torch.cuda.CUDAGraph`).

    The step is ~45 launches and ~1.6 ms of GPU work; issuing it eagerly costs 1.4-1.6 ms of host time (autograd
    engine, ctypes, optimizer bookkeeping), i.e. the host is as slow as the GPU.  libnerfhip launches on torch's
    current stream, allocates nothing and never synchronises, so the whole step captures; replay costs ~15 us of host
    time.  The first `warmup` calls run eagerly on the real batches (they are ordinary training steps), the next call
    captures and then replays.  A learning-rate change (scheduler) triggers a re-capture.

    With a `grad_sync` (N > 1 ranks) the step is by default ONE graph (round 6): the all-reduce the backward's grads-ready hook
    issues (parallel.GradSync: one 4.77 MB message over the step's joint gradient buffer; per model under form="per_model") is
    captured inside it (RCCL supports capture) between the reduce launch and Adam, and the host touches the step once.  When the
    capture of a collective fails on ANY rank the ranks agree (GradSync.agree_any, an eager collective outside every capture) to
    fall back to TWO graphs — [forward + backward] and [optimizer] — with the all-reduce issued eagerly in between: the form
    `sync_in_graph=False` / NERFHIP_SYNC_IN_GRAPH=0 selects outright.  World-1 A/B on one MI355X (RCCL communicator of one rank):
    profiles/README.md round 6.
    Outputs are static tensors overwritten by every replay (clone what you keep)."""

    def __init__(self, system, optimizer, grad_sync=None, warmup=3, sync_in_graph=None, backend=None, batch_source=None):
        self.system, self.opt, self.grad_sync = system, optimizer, grad_sync
        # `batch_source`: a callable returning the next {'rays', 'rgbs'} batch from device-resident data (RayStore.sample with
        # the default generator).  It is then called INSIDE the step, i.e. captured into the graph: every replay draws a fresh
        # batch (torch's graph-safe Philox state advances per replay) and no batch is copied into static buffers.
        self.batch_source = batch_source
        self._seed = None
        self.warmup = warmup
        # N > 1 ranks: ONE graph with the collective captured inside is the default (round 6); every rank that cannot capture it
        # makes all ranks fall back to two graphs with the all-reduce issued eagerly in between (_capture below), so the first
        # hardware run at N > 1 meets the fast form and still completes on a stack that refuses it.  NERFHIP_SYNC_IN_GRAPH=0 /
        # sync_in_graph=False selects the two-graph form outright.
        if sync_in_graph is None:
            sync_in_graph = os.environ.get("NERFHIP_SYNC_IN_GRAPH", "1") != "0"
        self.sync_in_graph = bool(sync_in_graph)
        self.calls = 0
        self.graph = None
        self.graph_opt = None
        self.static_batch = None
        self.static_out = None
        self.captured_lr = None
        self.capture_fallback = None      # repr of the exception that made the one-graph capture fall back to two graphs
        self._draw_state = None           # draws.GraphDrawState: the generator stream of the captured draw launches
        # `backend` abstracts the three device facilities the stepper needs (side stream, graph capture, replay) so the
        # host logic — warm-up, capture, two-graph step with the eager collective in between, re-capture on lr change —
        # can be exercised by the world-2 gloo CPU test with a recording stand-in (tests/test_distributed_cpu.py).
        self.backend = backend if backend is not None else _HipGraphBackend()

    @staticmethod
    def _detached(out):
        """The caller gets values, never the autograd graph (a kept-alive loss keeps the previous iteration's
        AccumulateGrad nodes alive across the capture)."""
        def d(v):
            return v.detach() if torch.is_tensor(v) else ({k: d(x) for k, x in v.items()} if isinstance(v, dict) else v)
        return d(out)

    def _fwd_bwd(self, batch):
        if self.batch_source is not None:
            batch = self.batch_source()
        out = self.system.training_step(batch, self.calls)
        self.opt.zero_grad(set_to_none=True)
        if self._seed is None:                 # created in an eager warm-up step, reused by every later (captured) one:
            from .ops import unit_seed                       # spares the ones_like fill of a bare .backward(), and the fused
            self._seed = unit_seed(out['loss']) if out['loss'].is_cuda else torch.ones_like(out['loss'])   # backward nodes
        out['loss'].backward(self._seed)                     # recognise it: no multiplication by d loss / d loss either
        return self._detached(out)

    def _eager(self, batch):
        out = self._fwd_bwd(batch)
        if self.grad_sync is not None:
            self.grad_sync.sync()
        self.opt.step()
        return out

    def _two_graphs(self):
        return self.grad_sync is not None and not self.sync_in_graph

    def _set_hooks(self, on):
        if self.grad_sync is not None and hasattr(self.grad_sync, "hooks_enabled"):
            self.grad_sync.hooks_enabled = on

    def _arm_draws(self):
        """The captured step's draw launches (draws.py) read their (seed, offset) from device memory: load it from torch's
        generator before the capture."""
        p = next(iter(self.system.parameters()), None)
        if p is None or not p.is_cuda:
            return
        from .draws import GraphDrawState
        if self._draw_state is None:
            self._draw_state = GraphDrawState(p.device)       # one per stepper: its graphs' launches hold this device buffer
        self._draw_state.arm()

    def _capture_with_draws(self, fn, **kw):
        if self._draw_state is None:
            return self.backend.capture(fn, **kw)
        from .draws import capturing
        with capturing(self._draw_state):
            return self.backend.capture(fn, **kw)

    def _capture(self, batch):
        self.static_batch = {k: (v.clone() if torch.is_tensor(v) else {kk: vv.clone() for kk, vv in v.items()})
                             for k, v in batch.items() if torch.is_tensor(v) or isinstance(v, dict)} if batch is not None else None
        self.captured_lr = get_learning_rate(self.opt)
        self._arm_draws()
        self.static_out = None
        self.graph_opt = None
        if not self._two_graphs():
            # one graph: forward, backward, [all-reduces issued from the grad-ready hooks, i.e. overlapping the rest of
            # the backward also on replay], optimizer
            err = None
            try:
                self.graph, self.static_out = self._capture_with_draws(lambda: self._eager(self.static_batch))
            except Exception as e:          # a failed stream capture / a collective that cannot be captured (RuntimeError), or
                if self.grad_sync is None:  # anything else one rank raises: its peers are about to enter agree_any and must not
                    raise                   # be left blocked in that collective
                err = e
            # A communicator whose collectives cannot be captured on this stack: keep the collectives outside the graphs
            # instead.  EVERY rank must take the same branch (one rank replaying a captured all-reduce while another issues an
            # eager one deadlocks), so the ranks agree on it with a collective of its own, outside any capture.
            failed = err is not None
            if self.grad_sync is not None and hasattr(self.grad_sync, "agree_any"):
                failed = self.grad_sync.agree_any(failed)
            if failed:
                import warnings
                warnings.warn("GraphedTrainStep: capturing the step with its all-reduces inside failed (%r); falling back to two "
                              "graphs with the collectives issued eagerly in between" % (err,))
                self.capture_fallback = repr(err) if err is not None else "another rank's capture failed"
                self.sync_in_graph = False
                self.graph = None
                getattr(self.grad_sync, "_inflight", {}).clear()
                self._arm_draws()           # the aborted capture recorded draw increments that will not be replayed
        if self._two_graphs():
            # two graphs with the collectives issued eagerly in between: the hooks must stay silent, or the fine
            # model's all-reduce would be captured into the first graph
            self._set_hooks(False)
            try:
                self.graph, self.static_out = self._capture_with_draws(lambda: self._fwd_bwd(self.static_batch))
                self.graph_opt, _ = self.backend.capture(self.opt.step, share_pool_with=self.graph)
            finally:
                self._set_hooks(True)

    def __call__(self, batch=None):
        if (batch is None) != (self.batch_source is not None):
            raise ValueError("pass a batch, or construct the stepper with a batch_source (not both)")
        self.calls += 1
        if self.calls <= self.warmup:
            return self.backend.on_side_stream(self._eager, batch)
        if (self.graph is None or get_learning_rate(self.opt) != self.captured_lr
                or (batch is not None and any(torch.is_tensor(batch[k]) and batch[k].shape != self.static_batch[k].shape for k in batch))):
            self._capture(batch)
        elif batch is not None:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self.static_batch[k].copy_(v, non_blocking=True)
                elif isinstance(v, dict):
                    for kk, vv in v.items():
                        self.static_batch[k][kk].copy_(vv, non_blocking=True)
        ds = self._draw_state
        if ds is not None and ds.increment:
            ds.before_replay()
        self.graph.replay()
        if ds is not None:
            ds.after_replay()
        if self.graph_opt is not None:
            self.grad_sync.sync()
            self.graph_opt.replay()
        return self.static_out


class _HipGraphBackend:
    """Side stream + hipGraph capture/replay through torch.cuda.CUDAGraph.

    Eager warm-up steps and the capture run on ONE side stream: autograd's AccumulateGrad nodes remember the stream they
    were created on, and a node created on the default stream while a later backward is being captured on another
    stream is undefined behaviour (observed: silently missing parameter updates, or a segfault)."""

    def __init__(self, keep_graph=False):
        self.stream = torch.cuda.Stream()
        self.keep_graph = keep_graph         # keep the captured hipGraph_t next to its executable (node_count)

    def on_side_stream(self, fn, *args):
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            out = fn(*args)
        torch.cuda.current_stream().wait_stream(self.stream)
        return out

    def capture(self, fn, share_pool_with=None):
        graph = torch.cuda.CUDAGraph(keep_graph=True) if self.keep_graph else torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        kw = {"pool": share_pool_with.pool()} if share_pool_with is not None else {}
        # thread_local: with a process group alive, RCCL's watchdog THREAD polls the events of earlier eager collectives; under
        # the default (global) capture mode one such hipEventQuery during the capture aborts the process ("operation not
        # permitted when stream is capturing") — a race that only shows on some runs
        with torch.cuda.graph(graph, stream=self.stream, capture_error_mode="thread_local", **kw):
            out = fn()
        return graph, out


def graph_node_count(graph):
    """Number of nodes (kernel launches, copies, ...) of a captured `torch.cuda.CUDAGraph(keep_graph=True)`: hipGraphGetNodes on
    its raw hipGraph_t.  None when the handle is not available."""
    import ctypes
    try:
        raw = graph.raw_cuda_graph()
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0:
            return None
        return int(n.value)
    except Exception:  # noqa: BLE001 - a diagnostic, never fatal
        return None
