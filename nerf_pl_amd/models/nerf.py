"""`models.nerf` of kwea123/nerf_pl, MI355X-native.

Same classes, constructor signatures, attributes and `state_dict` layout as the reference
(models/nerf.py:4-124) so checkpoints, optimizers and DDP see an ordinary nn.Module; `forward`
dispatches to the HIP kernels of libnerfhip (posenc / fused MFMA MLP) instead of ATen op chains.
"""
import torch
from torch import nn

from .. import default_mlp_dtype, ops


class Embedding(nn.Module):
    def __init__(self, in_channels, N_freqs, logscale=True):
        """Embeds x to (x, sin(2^k x), cos(2^k x), ...).  Reference: models/nerf.py:5-19."""
        super(Embedding, self).__init__()
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.funcs = [torch.sin, torch.cos]
        self.out_channels = in_channels * (len(self.funcs) * N_freqs + 1)
        if logscale:
            self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs)
        else:
            self.freq_bands = torch.linspace(1, 2 ** (N_freqs - 1), N_freqs)
        self.logscale = logscale

    def _device_bands(self, device):
        """`freq_bands` stays the plain CPU tensor the reference builds (nerf.py:16-19: not a buffer, not in the state_dict); its
        copy on `device` is made once and kept — a pageable H2D copy per forward would synchronise the stream and cannot be
        captured into a hipGraph."""
        cache = self.__dict__.setdefault("_bands_on", {})
        key = (device.type, device.index)
        hit = cache.get(key)
        if hit is None or hit[0] is not self.freq_bands:
            hit = cache[key] = (self.freq_bands, self.freq_bands.to(device, torch.float32).contiguous())
        return hit[1]

    def forward(self, x):
        """x: (B, in_channels) -> (B, out_channels).  Reference: models/nerf.py:21-38.
        One HIP launch (K1 posenc) instead of 41; channel order identical to the reference."""
        if x.shape[-1] != self.in_channels:
            raise ValueError("expected %d input channels" % self.in_channels)
        lead = x.shape[:-1]
        # logscale=True (the reference's only use, train.py:34-35): the kernel forms the bands 2^k itself; logscale=False: the
        # linspace bands of nerf.py:16-19 travel to the kernel as this module built them.  (The FUSED render_rays / NeRF path
        # encodes in-register with the logscale bands only: models/rendering._fusable.)
        bands = None if self.logscale else self._device_bands(x.device)
        out = ops.posenc(x.reshape(-1, self.in_channels).float(), self.N_freqs, bands=bands)
        return out.reshape(*lead, self.out_channels)


class NeRF(nn.Module):
    def __init__(self, D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4]):
        """Reference: models/nerf.py:42-81 (identical submodule names => identical state_dict keys)."""
        super(NeRF, self).__init__()
        self.D = D
        self.W = W
        self.in_channels_xyz = in_channels_xyz
        self.in_channels_dir = in_channels_dir
        self.skips = skips
        for i in range(D):
            if i == 0:
                layer = nn.Linear(in_channels_xyz, W)
            elif i in skips:
                layer = nn.Linear(W + in_channels_xyz, W)
            else:
                layer = nn.Linear(W, W)
            layer = nn.Sequential(layer, nn.ReLU(True))
            setattr(self, f"xyz_encoding_{i+1}", layer)
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), nn.ReLU(True))
        self.sigma = nn.Linear(W, 1)
        self.rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
        # --- MI355X specifics (not part of state_dict) ---
        self.mlp_dtype = default_mlp_dtype()    # 'fp32' (parity) | 'bf16' | 'bf16_f8' (roofline); a non-default shape runs
                                                # layer by layer (models/layered.py) and reads 'bf16_f8' as 'bf16'
        self._packed_cache = {}
        # weights replaced wholesale (load_ckpt, utils/__init__.py:55-76): weight images packed ahead of a step are stale
        self.register_load_state_dict_post_hook(lambda module, _incompatible: module._bump_weights_serial())

    def _bump_weights_serial(self):
        self._weights_serial = getattr(self, "_weights_serial", 0) + 1

    # -- fused-kernel plumbing -----------------------------------------------------------------
    def is_default_arch(self):
        return (self.D == 8 and self.W == 256 and self.in_channels_xyz == 63 and self.in_channels_dir == 27
                and list(self.skips) == [4])

    def linears(self):
        """The 12 nn.Linear modules in ops.PARAM_ORDER."""
        ls = [getattr(self, f"xyz_encoding_{i+1}")[0] for i in range(self.D)]
        return ls + [self.xyz_encoding_final, self.dir_encoding[0], self.sigma, self.rgb[0]]

    def flat_params(self):
        """[w0..w11, b0..b11] (the order autograd Functions take them in).  The list is cached (8 lookups per
        training step): nn.Parameter objects keep their identity across .to()/load_state_dict/optimizer steps."""
        fp = self.__dict__.get("_flat_params_list")
        if fp is None or fp[0] is not self.xyz_encoding_1[0].weight or fp[-1] is not self.rgb[0].bias:
            ls = self.linears()
            fp = [l.weight for l in ls] + [l.bias for l in ls]
            self.__dict__["_flat_params_list"] = fp
        return fp

    def _pack_args(self):
        """ctypes pointer tables of the 24 parameter tensors, validated once and rebuilt only when a parameter's
        storage moved (host-side cost matters: a training step re-packs four times and lasts ~1.7 ms)."""
        ps = self.flat_params()
        key = tuple(p.data_ptr() for p in ps)
        hit = self._packed_cache.get("args")
        if hit is None or hit[0] != key:
            hit = (key,) + ops.pack_arg_tables(ps[:12], ps[12:])
            self._packed_cache["args"] = hit
        return hit[1], hit[2], ps[0].device

    def packed_weights(self, dtype=None):
        """MFMA-fragment-ordered image of the CURRENT parameters (one ~5 us HIP launch into a reused buffer).

        Repacked on every call: parameter version counters are not a safe cache key — fused/foreach optimizers
        (`torch.optim.Adam(fused=True)`) update parameters without bumping `_version`, and a stale image would
        silently render/train with old weights.  Callers that know the weights are frozen (an eval loop) can hold
        on to the returned buffer."""
        if not self.is_default_arch():
            raise NotImplementedError("the fused HIP MLP implements the reference's default architecture "
                                      "(D=8, W=256, skips=[4], 63/27 inputs) only")
        dtype = dtype or self.mlp_dtype
        wp, bp, dev = self._pack_args()
        buf = self._packed_cache.get(dtype)
        if buf is None or buf.device != dev:
            buf = self._packed_cache[dtype] = torch.empty(ops.packed_bytes(dtype), device=dev, dtype=torch.uint8)
        ops.pack_weights_raw(wp, bp, buf, dtype)
        return buf

    def _bwd_buffer(self, dtype, dev):
        buf = self._packed_cache.get(("bwd", dtype))
        if buf is None or buf.device != dev:
            buf = self._packed_cache[("bwd", dtype)] = torch.empty(ops.packed_bwd_bytes(dtype), device=dev, dtype=torch.uint8)
        return buf

    def packed_weights_bwd(self, dtype=None):
        """W^T stream for the backward chain + the fp32 fold block (same policy as packed_weights)."""
        dtype = dtype or self.mlp_dtype
        wp, bp, dev = self._pack_args()
        buf = self._bwd_buffer(dtype, dev)
        ops.pack_weights_bwd_raw(wp, bp, buf, dtype)
        return buf

    def train_buffers(self, dtype, dev):
        """(forward image buffer, W^T image buffer) of this model, allocated once per (dtype, device)."""
        buf = self._packed_cache.get(dtype)
        if buf is None or buf.device != dev:
            buf = self._packed_cache[dtype] = torch.empty(ops.packed_bytes(dtype), device=dev, dtype=torch.uint8)
        return buf, self._bwd_buffer(dtype, dev)

    def packed_weights_train(self, dtype=None):
        """(forward image, W^T image) of the current parameters in ONE launch: a training forward packs both, its backward
        reuses the second.  The images live in ONE buffer per model: a later training forward of the same model re-packs them.
        That is harmless while the weights are unchanged (identical images) and WRONG if an optimizer stepped in between —
        PyTorch raises for its own saved tensors in that situation; here it is detected only for optimizers that announce their
        updates (FlatAdam bumps `_weights_serial`; `check_pack_serial` in the backward), not for foreign in-place updates."""
        if not self.is_default_arch():
            raise NotImplementedError("the fused HIP MLP implements the reference's default architecture "
                                      "(D=8, W=256, skips=[4], 63/27 inputs) only")
        dtype = dtype or self.mlp_dtype
        wp, bp, dev = self._pack_args()
        buf = self._packed_cache.get(dtype)
        if buf is None or buf.device != dev:
            buf = self._packed_cache[dtype] = torch.empty(ops.packed_bytes(dtype), device=dev, dtype=torch.uint8)
        bwd = self._bwd_buffer(dtype, dev)
        ops.pack_weights_train_raw(wp, bp, buf, bwd, dtype)
        self._packed_serial = getattr(self, "_weights_serial", 0)
        return buf, bwd

    def check_pack_serial(self, serial):
        """backward side of packed_weights_train: the W^T image must still be the one packed from the weights of `serial`"""
        if getattr(self, "_packed_serial", 0) != serial:
            raise RuntimeError("the W^T image this backward needs was re-packed from UPDATED weights by a later training forward of "
                               "the same model (an optimizer stepped between this graph's forward and its backward)")

    def forward(self, x, sigma_only=False):
        """x: (B, 63+27) embedded position+direction, or (B, 63) when sigma_only.
        Returns (B,4)=[rgb, sigma] or (B,1) sigma.  Reference: models/nerf.py:83-124."""
        if not self.is_default_arch():
            # any other D / W / skips / channel counts: one HIP GEMM launch per layer (models/layered.py, csrc/linear.hip)
            from .layered import nerf_forward
            lead = x.shape[:-1]
            out = nerf_forward(self, x.reshape(-1, x.shape[-1]), bool(sigma_only))
            return out.reshape(*lead, out.shape[-1])
        from .mlp_autograd import mlp_embedded
        return mlp_embedded(self, x, bool(sigma_only))
