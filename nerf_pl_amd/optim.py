"""Adam over flat parameter storage ("fused Adam for the 48 small tensors").

The reference builds `torch.optim.Adam(lr, eps=1e-8, weight_decay)` over every parameter of both models
(utils/__init__.py:10-30): 48 tensors, 1.19 M floats.  Here each model's 24 tensors become views of ONE flat fp32
buffer (in the order of `NeRF.flat_params()`, which is also the order of the flat gradient buffer the dW-reduce kernel
writes), and the update of all models is ONE hand-written HIP launch (`nerfhip_adam_step`, csrc/optim.hip) whose
`.grad` inputs are adopted from the HIP backward without a copy.  The step counter lives on the device, so a whole
training step replays as a hipGraph.

`state_dict()` / `load_state_dict()` speak the PER-PARAMETER layout of `torch.optim.Adam` over
`[p for m in models for p in m.parameters()]` (48 entries: `step`, `exp_avg`, `exp_avg_sq`), i.e. what the reference's
optimizer — and a Lightning checkpoint's `optimizer_states` (train.py `resume_from_checkpoint`) — holds, so optimizer
state moves between this class and the reference's Adam in both directions.
"""
import ctypes
import weakref

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


class FlatAdam(torch.optim.Optimizer):
    """Must be built AFTER the models' final device placement: `.to()/.cuda()/.float()` re-create parameter storage
    and break the aliasing with the flat buffers (`step()` verifies the aliasing and re-aliases or raises)."""

    def __init__(self, models, lr=5e-4, eps=1e-8, weight_decay=0, betas=(0.9, 0.999)):
        self.models = list(models)
        self.flats = []
        for m in self.models:
            self.flats.append(torch.nn.Parameter(self._flatten(m), requires_grad=True))
            # this optimizer announces every update (_bump_serial): weight images packed AHEAD of a step
            # (RayStore.sample(pack_models=...)) may be trusted while the serial stands — and only while THIS optimizer is alive: the
            # mark is a weak reference, so a model later stepped by another optimizer is not vouched for by a dead FlatAdam
            m._serial_tracked = weakref.ref(self)
        super().__init__(self.flats, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev = self.flats[0].device
        if not self.flats[0].is_cuda:
            raise _lib.NerfHipError("FlatAdam runs on MI355X only (no CPU fallback); use torch.optim.Adam on CPU tensors")
        self.exp_avg = [torch.zeros_like(f.data) for f in self.flats]
        self.exp_avg_sq = [torch.zeros_like(f.data) for f in self.flats]
        # [step count (float), arrival ticket (uint32 bits)] — device-resident: graph replays advance the counter
        self.dev_state = torch.zeros(2, device=dev, dtype=torch.float32)
        self._tables = None
        self._applied = None          # ids of the models whose update the last backward already applied (fused reduce + Adam)
        self._stepped = None          # indices of the models the first step updated (must stay the same: one shared step counter)

    # ---------------------------------------------------------------------------------------------- flat storage
    @staticmethod
    def _flatten(model):
        ps = model.flat_params()
        flat = torch.empty(sum(p.numel() for p in ps), device=ps[0].device, dtype=torch.float32)
        off = 0
        for p in ps:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)          # the module's parameters now alias the flat buffer
            off += n
        return flat

    def _check_alias(self):
        """Every module parameter must still be a view of its slice of the flat buffer (a later model.to()/.float()/
        p.data assignment silently breaks this: Adam would update the flat buffer while the kernels read the detached
        module parameters).  Re-alias when the detached parameters still have the right shape/device, else raise."""
        for m, flat in zip(self.models, self.flats):
            off = 0
            base = flat.data_ptr()
            broken = False
            for p in m.flat_params():
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    broken = True
                off += p.numel()
            if not broken:
                continue
            ps = m.flat_params()
            if any(p.device != flat.device for p in ps):
                raise _lib.NerfHipError("FlatAdam: model parameters moved to another device after the optimizer was built; "
                                        "rebuild the optimizer after the final .to()/.cuda()")
            off = 0
            with torch.no_grad():
                for p in ps:                                   # adopt the detached values, then alias again
                    n = p.numel()
                    flat.data[off:off + n].copy_(p.data.reshape(-1).float())
                    p.data = flat.data[off:off + n].view(p.shape)
                    off += n
            self._tables = None

    def _gather_grads(self):
        for m, flat in zip(self.models, self.flats):
            ps = m.flat_params()
            fg = getattr(m, "_flat_grad", None)
            g0, g1 = ps[0].grad, ps[-1].grad
            if (fg is not None and g0 is not None and g1 is not None and g0.data_ptr() == fg.data_ptr()
                    and g1.data_ptr() + g1.numel() * 4 == fg.data_ptr() + fg.numel() * 4):
                flat.grad = fg                                    # written in place by mlp_bwd_reduce: no copy
                # (first and last parameter's .grad alias the two ends of the flat buffer => autograd adopted the views)
            elif any(p.grad is not None for p in ps):
                g = torch.zeros_like(flat)
                off = 0
                for p in ps:
                    n = p.numel()
                    if p.grad is not None:
                        g[off:off + n].copy_(p.grad.reshape(-1))
                    off += n
                flat.grad = g
            else:
                flat.grad = None

    # ---------------------------------------------------------------------------------------------- update inside the backward
    def handle(self, models):
        """nerfhip_adam_fused for `models` (in that order): the fused training step's reduce kernel applies this optimizer's
        update to the flat parameter storage while it writes the gradients (models/train_step.py; single-GPU steps only —
        with several ranks the all-reduce sits between gradients and update)."""
        self._check_alias()
        h = _lib.AdamFused()
        h.n_models = len(models)
        for k, m in enumerate(models):
            i = next(j for j, mm in enumerate(self.models) if mm is m)
            h.param[k] = self.flats[i].data_ptr()
            h.exp_avg[k] = self.exp_avg[i].data_ptr()
            h.exp_avg_sq[k] = self.exp_avg_sq[i].data_ptr()
        g = self.param_groups[0]
        h.state = self.dev_state.data_ptr()
        h.lr, h.beta1, h.beta2, h.eps, h.weight_decay = float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), \
            float(g['weight_decay'])
        return h

    def applied_in_backward(self, models):
        self._applied = {id(m) for m in models}
        self._bump_serial([j for j, mm in enumerate(self.models) if any(mm is m for m in models)])

    def _bump_serial(self, idx):
        """the weights of these models changed: packed weight images made before this point are stale (models/nerf.py)"""
        for i in idx:
            self.models[i]._weights_serial = getattr(self.models[i], "_weights_serial", 0) + 1

    # ---------------------------------------------------------------------------------------------- the update
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._applied is not None:
            # the backward that produced these gradients already updated the parameters (and advanced the step counter)
            done, self._applied = self._applied, None
            if done != {id(m) for m in self.models}:
                raise _lib.NerfHipError("FlatAdam: the fused backward updated only some of the optimizer's models")
            if self._stepped is None:
                self._stepped = tuple(range(len(self.models)))
            return loss
        self._check_alias()
        self._gather_grads()
        idx = [i for i, f in enumerate(self.flats) if f.grad is not None]
        if not idx:
            return loss
        # ONE device-resident step counter serves every model (torch.optim.Adam keeps one per parameter): that is only the same
        # thing while the same models are stepped every time — a model that sat out a step would get the others' bias correction
        if self._stepped is None:
            self._stepped = tuple(idx)
        elif self._stepped != tuple(idx):
            raise _lib.NerfHipError("FlatAdam: the set of models with gradients changed between steps (%s -> %s); one shared step "
                                    "counter cannot represent that — use torch.optim.Adam for partially frozen training"
                                    % (self._stepped, tuple(idx)))
        self._bump_serial(idx)
        g = self.param_groups[0]
        n = len(idx)
        arr = ctypes.c_void_p * n
        pp = arr(*[self.flats[i].data_ptr() for i in idx])
        gp = arr(*[self.flats[i].grad.data_ptr() for i in idx])
        mp = arr(*[self.exp_avg[i].data_ptr() for i in idx])
        vp = arr(*[self.exp_avg_sq[i].data_ptr() for i in idx])
        nn = (ctypes.c_int64 * n)(*[self.flats[i].numel() for i in idx])
        with torch.cuda.device(self.flats[0].device):
            check(_lib.load().nerfhip_adam_step(pp, gp, mp, vp, nn, n, ptr(self.dev_state), float(g['lr']), float(g['betas'][0]),
                                                float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), stream_ptr()),
                  "nerfhip_adam_step")
        return loss

    def zero_grad(self, set_to_none=True):
        self._applied = None
        for m, flat in zip(self.models, self.flats):
            flat.grad = None
            m._flat_grad = None
            for p in m.flat_params():
                if set_to_none or p.grad is None:
                    p.grad = None
                else:
                    p.grad.zero_()

    # ---------------------------------------------------------------------------------------------- checkpoints
    def _param_slices(self):
        """(model index, flat offset, shape) of every parameter in `[p for m in models for p in m.parameters()]` order —
        the order the reference's optimizer (utils/__init__.py:12-14) numbers its parameters in."""
        out = []
        for mi, m in enumerate(self.models):
            offs, off = {}, 0
            for p in m.flat_params():
                offs[id(p)] = off
                off += p.numel()
            for p in m.parameters():
                out.append((mi, offs[id(p)], tuple(p.shape)))
        return out

    def state_dict(self):
        """torch.optim.Adam's layout over the 48 module parameters (loadable by the reference's optimizer)."""
        sl = self._param_slices()
        step = self.dev_state[0].detach().clone()
        state = {}
        if float(step) > 0:
            for i, (mi, off, shape) in enumerate(sl):
                n = 1
                for s in shape:
                    n *= s
                state[i] = {"step": step.clone().cpu(),
                            "exp_avg": self.exp_avg[mi][off:off + n].view(shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[mi][off:off + n].view(shape).clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"],
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False, "params": list(range(len(sl)))}
        if "initial_lr" in g:
            group["initial_lr"] = g["initial_lr"]
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        sl = self._param_slices()
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(sl):
            raise ValueError("FlatAdam.load_state_dict expects torch.optim.Adam state over the %d model parameters" % len(sl))
        g = self.param_groups[0]
        for k in ("lr", "eps", "weight_decay", "initial_lr"):
            if k in groups[0]:
                g[k] = groups[0][k]
        if "betas" in groups[0]:
            g["betas"] = tuple(groups[0]["betas"])
        ids = groups[0]["params"]
        step = 0.0
        for e in self.exp_avg + self.exp_avg_sq:
            e.zero_()
        for i, (mi, off, shape) in enumerate(sl):
            st = sd["state"].get(ids[i])
            if st is None:
                continue
            n = st["exp_avg"].numel()
            self.exp_avg[mi][off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[mi][off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            step = max(step, float(st["step"]))
        self.dev_state.zero_()
        self.dev_state[0] = step
