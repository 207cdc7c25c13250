"""Adam over flat parameter storage ("fused Adam for the 48 small tensors").

The reference builds `torch.optim.Adam(lr, eps=1e-8, weight_decay)` over every parameter of both models
(utils/__init__.py:10-30): 48 tensors, 1.19 M floats.  At a ~1.8 ms training step the multi-tensor Adam launches
(~100 us) are 5 % of the step.  Here each model's 24 tensors become views of ONE flat fp32 buffer (in the order of
`NeRF.flat_params()`, which is also the order of the flat gradient buffer the dW-reduce kernel writes), so the
update is the same Adam arithmetic on 2 tensors whose `.grad` is adopted from the HIP backward without a copy.
`state_dict` keys of the models are unchanged (the nn.Parameters still exist, they just alias the flat storage).
"""
import torch


class FlatAdam(torch.optim.Adam):
    def __init__(self, models, lr=5e-4, eps=1e-8, weight_decay=0, betas=(0.9, 0.999)):
        self.models = list(models)
        self.flats = []
        for m in self.models:
            ps = m.flat_params()
            dev = ps[0].device
            flat = torch.empty(sum(p.numel() for p in ps), device=dev, dtype=torch.float32)
            off = 0
            for p in ps:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view(p.shape)          # the module's parameters now alias the flat buffer
                off += n
            self.flats.append(torch.nn.Parameter(flat))
        on_gpu = self.flats[0].is_cuda
        # capturable: the step counters live on the device, so a whole training step can be replayed as a hipGraph
        super().__init__(self.flats, lr=lr, eps=eps, weight_decay=weight_decay, betas=betas, fused=on_gpu, capturable=on_gpu)

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

    @torch.no_grad()
    def step(self, closure=None):
        self._gather_grads()
        return super().step(closure)

    def zero_grad(self, set_to_none=True):
        for m, flat in zip(self.models, self.flats):
            flat.grad = None
            m._flat_grad = None
            for p in m.flat_params():
                if set_to_none or p.grad is None:
                    p.grad = None
                else:
                    p.grad.zero_()
