"""Python-side operators over libnerfhip's C ABI (include/nerfhip.h).

Each function allocates its outputs with torch (torch owns all HBM), passes raw device pointers +
the current HIP stream through ctypes, and raises `NerfHipError` on any non-zero return.  There is
no CPU path and no eager-PyTorch fallback: tensors must live on an MI355X.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import BF16, BF16_F8, F32, NerfHipError, check, device_guard, ptr, require_gpu, stream_ptr

# 'bf16_f8': bf16 MFMA forward and dX chain; the tensors saved for the weight-gradient GEMM (activations, dY) are stored
# as block-scaled e4m3 and consumed by the MX-scaled fp8 MFMA (include/nerfhip.h: NERFHIP_BF16_F8)
_DTYPES = {"fp32": F32, "f32": F32, "float32": F32, torch.float32: F32, F32: F32,
           "bf16": BF16, "bfloat16": BF16, torch.bfloat16: BF16, "bf16_f8": BF16_F8}


def mlp_dtype_code(d):
    if isinstance(d, bool) or d not in _DTYPES:
        raise ValueError("mlp dtype must be 'fp32', 'bf16' or 'bf16_f8', got %r" % (d,))
    return _DTYPES[d]


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------- posenc (a2)
class _PosEnc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_freqs, bands):
        require_gpu(x, bands)
        x = _c(x)
        n, C = x.shape
        out = torch.empty(n, C * (2 * n_freqs + 1), device=x.device, dtype=torch.float32)
        check(_lib.load().nerfhip_posenc_bands(ptr(x), ptr(bands), ptr(out), n, C, n_freqs, stream_ptr()), "nerfhip_posenc")
        ctx.save_for_backward(x)
        ctx.n_freqs, ctx.bands = n_freqs, bands
        return out

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        gout = _c(gout.float())
        gx = torch.empty_like(x)
        n, C = x.shape
        check(_lib.load().nerfhip_posenc_bands_bwd(ptr(x), ptr(ctx.bands), ptr(gout), ptr(gx), n, C, ctx.n_freqs, stream_ptr()),
              "nerfhip_posenc_bwd")
        return gx, None, None


@device_guard
def posenc(x, n_freqs, bands=None):
    """Embedding.forward (reference models/nerf.py:21-38). x (n,C) -> (n, C(2F+1)).
    bands None: the logscale bands 2^k; else a device tensor of n_freqs frequency bands (`logscale=False`, nerf.py:16-19)."""
    if x.dim() != 2:
        raise ValueError("posenc expects (n, C)")
    if bands is not None:
        if bands.numel() != int(n_freqs):
            raise ValueError("posenc: need one band per frequency")
        bands = bands.to(x.device, torch.float32).contiguous()
    return _PosEnc.apply(x, int(n_freqs), bands)


# ------------------------------------------------------------------------------- sampling (a5, a8-a10)
@device_guard
def sample_coarse_z(rays, n_samples, use_disp=False, perturb=0.0, perturb_rand=None):
    """rendering.py:183-204.  rays (B,8) -> z (B,S)."""
    require_gpu(rays, perturb_rand)
    rays = _c(rays)
    B = rays.shape[0]
    z = torch.empty(B, n_samples, device=rays.device, dtype=torch.float32)
    if perturb > 0:
        if perturb_rand is None:
            raise ValueError("perturb>0 needs perturb_rand")
        perturb_rand = _c(perturb_rand)
    check(_lib.load().nerfhip_sample_coarse_z(ptr(rays), ptr(perturb_rand if perturb > 0 else None), ptr(z), B,
                                              n_samples, int(bool(use_disp)), float(perturb), stream_ptr()),
          "nerfhip_sample_coarse_z")
    return z


@device_guard
def searchsorted(a, v, out=None, side="left"):
    """Drop-in for torchsearchsorted.searchsorted (reference models/rendering.py:2,42): batched
    row-wise numpy-style searchsorted; a (B,M), v (B,K) float32 -> int64 (B,K).  Like the extension, `a` or `v` may have ONE
    row, which then serves every row of the other (the reference itself never uses that form: rendering.py:42)."""
    require_gpu(a, v)
    if side not in ("left", "right"):
        raise ValueError("side must be 'left' or 'right'")
    if a.dim() != 2 or v.dim() != 2 or not (a.shape[0] == v.shape[0] or a.shape[0] == 1 or v.shape[0] == 1):
        raise ValueError("searchsorted expects a (B,M) and v (B,K), or one of them with a single row")
    rows = max(a.shape[0], v.shape[0])
    a, v = _c(a.expand(rows, a.shape[1])), _c(v.expand(rows, v.shape[1]))
    B, M = a.shape
    K = v.shape[1]
    if out is None:
        out = torch.empty(B, K, device=a.device, dtype=torch.int64)
    elif out.dtype != torch.int64 or not out.is_contiguous() or tuple(out.shape) != (B, K):
        raise ValueError("out must be a contiguous int64 (B,K) tensor")
    fn = _lib.load().nerfhip_searchsorted_right if side == "right" else _lib.load().nerfhip_searchsorted_left
    check(fn(ptr(a), ptr(v), ptr(out), B, M, K, stream_ptr()), "nerfhip_searchsorted_" + side)
    return out


# Rounding of sample_pdf's normaliser `torch.sum(weights, -1)` (rendering.py:30; include/nerfhip.h NERFHIP_ROW_TOTAL_*):
#   "aten"  (default) the reference's own bits: ATen's fp32 addition order on a CPU.  The searchsorted indices of rendering.py:42
#           have knife edges on the last bit of this total; under this mode the (cdf, u) -> inds triples recorded at the
#           reference's call site are reproduced on EVERY element, and the fine model's gradients agree with the reference's to
#           ~1e-6 of each tensor's maximum instead of ~1e-2 (0.1-0.5 % of the fine samples land in another bin otherwise:
#           tests/test_gpu_training.py).  Costs ~30 dependent fp32 additions per ray.
#   "exact" the correctly rounded fp32 sum (host-independent); what the plain C entry points nerfhip_sample_pdf / nerfhip_fine_z use.
# Process-wide (every sample_pdf / fine_z / fused training launch reads it at call time); NERFHIP_ROW_TOTAL=exact in the environment.
_ROW_TOTAL_MODES = {"exact": 0, "aten": 1}
_row_total = _ROW_TOTAL_MODES[os.environ.get("NERFHIP_ROW_TOTAL", "aten")]


def set_row_total(mode):
    """Select the rounding of sample_pdf's row total: "exact" | "aten".  Returns the previous mode's name."""
    global _row_total
    prev = [k for k, v in _ROW_TOTAL_MODES.items() if v == _row_total][0]
    _row_total = _ROW_TOTAL_MODES[mode]
    return prev


@device_guard
def sample_pdf_u(bins, weights, n_importance, u=None, eps=1e-5, return_cdf_inds=False):
    """sample_pdf with explicit uniforms: u None -> deterministic linspace; (K,) or (B,K) otherwise.
    return_cdf_inds: also return the kernel's cdf (B,M+1) and searchsorted indices (B,K) int64 (rendering.py:31-42)."""
    require_gpu(bins, weights, u)
    if bins.stride(-1) != 1:
        bins = bins.contiguous()
    if weights.stride(-1) != 1:
        weights = weights.contiguous()
    B, M = weights.shape
    if bins.shape != (B, M + 1):
        raise ValueError("bins must be (N_rays, N_samples_+1)")
    u_stride = 0
    if u is not None:
        u = _c(u)
        if u.dim() == 2:
            if u.shape != (B, n_importance):
                raise ValueError("u must be (B, N_importance)")
            u_stride = n_importance
        elif u.shape != (n_importance,):
            raise ValueError("u must be (N_importance,) or (B, N_importance)")
    samples = torch.empty(B, n_importance, device=bins.device, dtype=torch.float32)
    cdf = torch.empty(B, M + 1, device=bins.device, dtype=torch.float32) if return_cdf_inds else None
    inds = torch.empty(B, n_importance, device=bins.device, dtype=torch.int64) if return_cdf_inds else None
    check(_lib.load().nerfhip_sample_pdf_ex(ptr(bins), bins.stride(0), ptr(weights), weights.stride(0), ptr(u), u_stride,
                                            ptr(samples), B, M, n_importance, float(eps), ptr(cdf), ptr(inds), _row_total,
                                            stream_ptr()),
          "nerfhip_sample_pdf")
    return (samples, cdf, inds) if return_cdf_inds else samples


@device_guard
def fine_z(z_coarse, w_coarse, n_importance, u=None, eps=1e-5, return_new=False, return_cdf_inds=False):
    """rendering.py:223-229 in one launch: z_fine = sort(cat(z_coarse, sample_pdf(z_mid, w[:,1:-1]))).
    return_cdf_inds: append the fused kernel's cdf (B,S-1) and searchsorted indices (B,N_i) to the result."""
    require_gpu(z_coarse, w_coarse, u)
    z_coarse, w_coarse = _c(z_coarse), _c(w_coarse)
    B, S = z_coarse.shape
    u_stride = 0
    if u is not None:
        u = _c(u)
        u_stride = n_importance if u.dim() == 2 else 0
    zf = torch.empty(B, S + n_importance, device=z_coarse.device, dtype=torch.float32)
    zn = torch.empty(B, n_importance, device=z_coarse.device, dtype=torch.float32) if return_new else None
    cdf = torch.empty(B, S - 1, device=z_coarse.device, dtype=torch.float32) if return_cdf_inds else None
    inds = torch.empty(B, n_importance, device=z_coarse.device, dtype=torch.int64) if return_cdf_inds else None
    check(_lib.load().nerfhip_fine_z_ex(ptr(z_coarse), ptr(w_coarse), ptr(u), u_stride, ptr(zf), ptr(zn), B, S,
                                        n_importance, float(eps), ptr(cdf), ptr(inds), _row_total, stream_ptr()), "nerfhip_fine_z")
    out = (zf, zn) if return_new else (zf,)
    if return_cdf_inds:
        out = out + (cdf, inds)
    return out if len(out) > 1 else out[0]


# ------------------------------------------------------------------------------- compositing (a7)
class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z, rays, noise, noise_std, white_back):
        require_gpu(raw, z, rays, noise)
        raw, z, rays = _c(raw), _c(z), _c(rays)
        B, S = z.shape
        raw_ch = 4 if raw.dim() == 3 else 1
        if raw.numel() != B * S * raw_ch:
            raise ValueError("raw must be (B,S,4) or (B,S)")
        if noise is not None:
            noise = _c(noise)
        dev = z.device
        weights = torch.empty(B, S, device=dev, dtype=torch.float32)
        opacity = torch.empty(B, device=dev, dtype=torch.float32)
        rgb = torch.empty(B, 3, device=dev, dtype=torch.float32) if raw_ch == 4 else None
        depth = torch.empty(B, device=dev, dtype=torch.float32) if raw_ch == 4 else None
        check(_lib.load().nerfhip_composite_fwd(ptr(raw), raw_ch, ptr(z), ptr(rays), ptr(noise), float(noise_std),
                                                int(bool(white_back)), ptr(weights), ptr(rgb), ptr(depth),
                                                ptr(opacity), B, S, stream_ptr()), "nerfhip_composite_fwd")
        ctx.save_for_backward(raw, z, rays, noise)
        ctx.cfg = (raw_ch, float(noise_std), int(bool(white_back)))
        ctx.set_materialize_grads(False)     # unused outputs (weights/depth/opacity) arrive as None, not as zero fills
        if raw_ch == 4:
            return weights, opacity, rgb, depth
        return weights, opacity

    @staticmethod
    def backward(ctx, g_weights, g_opacity, g_rgb=None, g_depth=None):
        raw, z, rays, noise = ctx.saved_tensors
        raw_ch, noise_std, white_back = ctx.cfg
        B, S = z.shape

        def prep(g):
            return None if g is None else _c(g.float())

        g_raw = torch.empty_like(raw)
        check(_lib.load().nerfhip_composite_bwd(ptr(raw), raw_ch, ptr(z), ptr(rays), ptr(noise), noise_std, white_back,
                                                ptr(prep(g_rgb)), ptr(prep(g_depth)), ptr(prep(g_opacity)),
                                                ptr(prep(g_weights)), ptr(g_raw), B, S, stream_ptr()),
              "nerfhip_composite_bwd")
        return g_raw, None, None, None, None, None


@device_guard
def composite(raw, z, rays, noise=None, noise_std=0.0, white_back=False):
    """Volume-rendering quadrature (rendering.py:143-172).
    raw (B,S,4) -> (weights, opacity, rgb, depth);  raw (B,S) sigma-only -> (weights, opacity)."""
    if noise_std == 0:
        noise = None
    return _Composite.apply(raw, z, rays, noise, float(noise_std), bool(white_back))


@device_guard
def composite_train(raw, z, rays, noise, noise_std, white_back, target, grad_scale, want_weights=True):
    """composite (raw (B,S,4)) + d MSE / d rgb against `target` (B,3) + the compositing backward for it, one launch
    (nerfhip_composite_train).  Returns (weights | None, opacity, rgb, depth, g_raw = d loss / d raw); nothing is recorded for
    autograd — the caller (models/train_step.py) owns the backward."""
    require_gpu(raw, z, rays, noise, target)
    raw, z, rays, target = _c(raw), _c(z), _c(rays), _c(target)
    B, S = z.shape
    if raw.numel() != B * S * 4 or target.numel() != B * 3:
        raise ValueError("composite_train: raw must be (B,S,4) and target (B,3)")
    if noise_std == 0:
        noise = None
    elif noise is not None:
        noise = _c(noise)
    dev = z.device
    weights = torch.empty(B, S, device=dev, dtype=torch.float32) if want_weights else None
    opacity = torch.empty(B, device=dev, dtype=torch.float32)
    rgb = torch.empty(B, 3, device=dev, dtype=torch.float32)
    depth = torch.empty(B, device=dev, dtype=torch.float32)
    g_raw = torch.empty(B, S, 4, device=dev, dtype=torch.float32)
    check(_lib.load().nerfhip_composite_train(ptr(raw), ptr(z), ptr(rays), ptr(noise), float(noise_std), int(bool(white_back)),
                                              ptr(target), float(grad_scale), ptr(weights), ptr(rgb), ptr(depth), ptr(opacity),
                                              ptr(g_raw), B, S, stream_ptr()), "nerfhip_composite_train")
    return weights, opacity, rgb, depth, g_raw


@device_guard
def composite_train_fine_z(raw, z, rays, noise, noise_std, white_back, target, grad_scale, n_importance, u=None, eps=1e-5,
                           want_weights=False):
    """The coarse pass of a training step in one launch (nerfhip_composite_train_fine_z): composite_train + fine_z on the
    ray's weights (which stay in LDS unless want_weights).  Returns (weights | None, opacity, rgb, depth, g_raw, z_fine)."""
    require_gpu(raw, z, rays, noise, target, u)
    raw, z, rays, target = _c(raw), _c(z), _c(rays), _c(target)
    B, S = z.shape
    if raw.numel() != B * S * 4 or target.numel() != B * 3:
        raise ValueError("composite_train_fine_z: raw must be (B,S,4) and target (B,3)")
    noise = None if noise_std == 0 else (_c(noise) if noise is not None else None)
    _check_draw("composite_train_fine_z", "noise", noise, B * S)
    u_stride = 0
    if u is not None:
        u = _c(u)
        u_stride = n_importance if u.dim() == 2 else 0
        _check_draw("composite_train_fine_z", "u", u, B * n_importance if u.dim() == 2 else n_importance)
    dev = z.device
    weights = torch.empty(B, S, device=dev, dtype=torch.float32) if want_weights else None
    opacity = torch.empty(B, device=dev, dtype=torch.float32)
    rgb = torch.empty(B, 3, device=dev, dtype=torch.float32)
    depth = torch.empty(B, device=dev, dtype=torch.float32)
    g_raw = torch.empty(B, S, 4, device=dev, dtype=torch.float32)
    zf = torch.empty(B, S + n_importance, device=dev, dtype=torch.float32)
    check(_lib.load().nerfhip_composite_train_fine_z(ptr(raw), ptr(z), ptr(rays), ptr(noise), float(noise_std), int(bool(white_back)),
                                                     ptr(target), float(grad_scale), ptr(weights), ptr(rgb), ptr(depth), ptr(opacity),
                                                     ptr(g_raw), B, S, ptr(u), u_stride, int(n_importance), float(eps), ptr(zf),
                                                     _row_total, stream_ptr()), "nerfhip_composite_train_fine_z")
    return weights, opacity, rgb, depth, g_raw, zf


_TICKETS = {}
_CAPTURES = {}
_EAGER_SLOTS = {}
_TICKET_WORDS = 1024


def _check_draw(op, name, t, numel):
    """caller-supplied draw tensors (batch['draws'] may have been made for other hyper-parameters): a wrong size would be an
    out-of-bounds device read, not an error"""
    if t is not None and (t.numel() != numel or t.dtype != torch.float32):
        raise ValueError("%s: %s must hold %d fp32 draws, got %s %s" % (op, name, numel, tuple(t.shape), t.dtype))


def _ticket(device):
    """One zero-initialised device word for the arrival-ticket kernels (they leave it at zero).  Eager launches on one stream are
    ordered and share the word of their (GPU, stream).  A hipGraph is different: it is replayed on whatever stream is current then,
    so two graphs CAPTURED on one stream may run concurrently — every capture therefore gets a word of its own: a new
    capture is recognised by the stream's capture status going from "none" to "active" between two calls."""
    st = torch.cuda.current_stream(device)
    capturing = torch.cuda.is_current_stream_capturing()
    rec = _CAPTURES.setdefault((device.index, st.stream_id), [False, 0])
    if capturing and not rec[0]:
        rec[1] += 1
    rec[0] = capturing
    pool = _TICKETS.get(device.index)
    if pool is None:
        if capturing:
            raise NerfHipError("the ticket words of this GPU must exist before a capture: run the step once eagerly first")
        pool = _TICKETS[device.index] = torch.zeros(_TICKET_WORDS * 4, device=device, dtype=torch.int32)
    # slots 0..31: eager use, one per stream in order of first use; 32..: captures, in order (16-byte pitch)
    if capturing:
        slot = 32 + rec[1] % (_TICKET_WORDS - 32)
    else:
        slot = _EAGER_SLOTS.setdefault((device.index, st.stream_id), len([k for k in _EAGER_SLOTS if k[0] == device.index]) % 32)
    return pool[4 * slot:4 * slot + 4]


@device_guard
def composite_train_loss(raw, z, rays, noise, noise_std, white_back, target, grad_scale, rgb_coarse=None):
    """The last pass of a training step in one launch (nerfhip_composite_train_loss): composite_train + the values of mse_psnr
    over (rgb_coarse, this pass's rgb).  Returns (opacity, rgb, depth, g_raw, out3 = [loss, psnr, mse])."""
    require_gpu(raw, z, rays, noise, target, rgb_coarse)
    raw, z, rays, target = _c(raw), _c(z), _c(rays), _c(target)
    B, S = z.shape
    if raw.numel() != B * S * 4 or target.numel() != B * 3 or (rgb_coarse is not None and rgb_coarse.numel() != B * 3):
        raise ValueError("composite_train_loss: raw must be (B,S,4), target and rgb_coarse (B,3)")
    noise = None if noise_std == 0 else (_c(noise) if noise is not None else None)
    _check_draw("composite_train_loss", "noise", noise, B * S)
    dev = z.device
    opacity = torch.empty(B, device=dev, dtype=torch.float32)
    rgb = torch.empty(B, 3, device=dev, dtype=torch.float32)
    depth = torch.empty(B, device=dev, dtype=torch.float32)
    g_raw = torch.empty(B, S, 4, device=dev, dtype=torch.float32)
    out3 = torch.empty(3, device=dev, dtype=torch.float32)
    check(_lib.load().nerfhip_composite_train_loss(ptr(raw), ptr(z), ptr(rays), ptr(noise), float(noise_std), int(bool(white_back)),
                                                   ptr(target), float(grad_scale), None, ptr(rgb), ptr(depth), ptr(opacity), ptr(g_raw),
                                                   B, S, ptr(_c(rgb_coarse)) if rgb_coarse is not None else None, ptr(out3),
                                                   ptr(_ticket(dev)), stream_ptr()), "nerfhip_composite_train_loss")
    return opacity, rgb, depth, g_raw, out3


# ------------------------------------------------------------------------------- loss + PSNR (N2)
class _MsePsnr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_c, rgb_f, target):
        require_gpu(rgb_c, rgb_f, target)
        rgb_c, target = _c(rgb_c), _c(target)
        rgb_f = _c(rgb_f) if rgb_f is not None else None
        n = target.numel()
        if rgb_c.numel() != n or (rgb_f is not None and rgb_f.numel() != n):
            raise ValueError("mse_psnr: shape mismatch")
        out3 = torch.empty(3, device=target.device, dtype=torch.float32)
        g_c = torch.empty_like(rgb_c)
        g_f = torch.empty_like(rgb_f) if rgb_f is not None else None
        check(_lib.load().nerfhip_mse_psnr(ptr(rgb_c), ptr(rgb_f), ptr(target), n, ptr(out3), ptr(g_c), ptr(g_f), stream_ptr()),
              "nerfhip_mse_psnr")
        ctx.save_for_backward(g_c, g_f)
        ctx.mark_non_differentiable(out3)
        ctx.set_materialize_grads(False)      # no zeros(3) launch for the unused gradient of out3
        return out3[0], out3

    @staticmethod
    def backward(ctx, g_loss, _g_out3):
        g_c, g_f = ctx.saved_tensors
        if g_loss is None:
            return None, None, None
        if is_unit_seed(g_loss):              # d loss / d loss == 1 (loss.backward(unit_seed(loss))): no scaling launch
            return g_c, g_f, None
        if g_f is None:
            return g_c * g_loss, None, None
        g_c, g_f = torch._foreach_mul([g_c, g_f], g_loss)      # one launch for both images
        return g_c, g_f, None


_UNIT_SEEDS = {}


def unit_seed(like):
    """A cached scalar 1.0 on `like`'s device to pass as `loss.backward(unit_seed(loss))`: the fused backward nodes recognise it
    (by storage) and skip the multiplication by d loss / d loss.  Never modify it in place."""
    key = (like.device.type, like.device.index)
    t = _UNIT_SEEDS.get(key)
    if t is None:
        t = _UNIT_SEEDS[key] = torch.ones((), device=like.device, dtype=torch.float32)
    return t


def is_unit_seed(g):
    t = _UNIT_SEEDS.get((g.device.type, g.device.index))
    return t is not None and g.data_ptr() == t.data_ptr() and g.numel() == 1


@device_guard
def mse_psnr_values(rgb_coarse, rgb_fine, target):
    """[loss, psnr, mse] (losses.py:9-14 + metrics.py:4-13) without gradients: the value half of nerfhip_mse_psnr."""
    require_gpu(rgb_coarse, rgb_fine, target)
    rgb_coarse, target = _c(rgb_coarse), _c(target)
    rgb_fine = _c(rgb_fine) if rgb_fine is not None else None
    out3 = torch.empty(3, device=target.device, dtype=torch.float32)
    check(_lib.load().nerfhip_mse_psnr(ptr(rgb_coarse), ptr(rgb_fine), ptr(target), target.numel(), ptr(out3), None, None, stream_ptr()),
          "nerfhip_mse_psnr")
    return out3


@device_guard
def mse_psnr(rgb_coarse, rgb_fine, target):
    """MSELoss.forward (losses.py:9-14) + psnr (metrics.py:4-13) + d loss/d rgb in one launch.
    Returns (loss [differentiable scalar], out3 = [loss, psnr, mse] detached)."""
    return _MsePsnr.apply(rgb_coarse, rgb_fine, target)


# ------------------------------------------------------------------------------- MLP (a3, a4, a6)
PARAM_ORDER = ["xyz_encoding_1.0", "xyz_encoding_2.0", "xyz_encoding_3.0", "xyz_encoding_4.0",
               "xyz_encoding_5.0", "xyz_encoding_6.0", "xyz_encoding_7.0", "xyz_encoding_8.0",
               "xyz_encoding_final", "dir_encoding.0", "sigma", "rgb.0"]
PARAM_SHAPES = [(256, 63), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256), (256, 256), (256, 256),
                (256, 256), (128, 283), (1, 256), (3, 128)]


def packed_bytes(dtype):
    return int(_lib.load().nerfhip_mlp_packed_bytes(mlp_dtype_code(dtype)))


@device_guard
def pack_weights(weights, biases, dtype, out=None):
    """Repack 12 (weight, bias) fp32 tensors (state_dict order, PARAM_ORDER) into the MFMA A-fragment
    stream of `dtype`.  Returns a uint8 device buffer."""
    code = mlp_dtype_code(dtype)
    if len(weights) != 12 or len(biases) != 12:
        raise ValueError("need 12 weights and 12 biases")
    keep = []
    for i, (w, b) in enumerate(zip(weights, biases)):
        require_gpu(w, b)
        if tuple(w.shape) != PARAM_SHAPES[i] or tuple(b.shape) != (PARAM_SHAPES[i][0],):
            raise NerfHipError("fused MLP supports the reference default architecture only "
                               "(D=8, W=256, skips=[4], in 63/27); got %s for %s" % (tuple(w.shape), PARAM_ORDER[i]))
        keep.append((_c(w.detach()), _c(b.detach())))
    dev = keep[0][0].device
    if out is None:
        out = torch.empty(packed_bytes(dtype), device=dev, dtype=torch.uint8)
    wp = (ctypes.c_void_p * 12)(*[k[0].data_ptr() for k in keep])
    bp = (ctypes.c_void_p * 12)(*[k[1].data_ptr() for k in keep])
    check(_lib.load().nerfhip_mlp_pack_weights(wp, bp, ptr(out), code, stream_ptr()), "nerfhip_mlp_pack_weights")
    return out


_SIZE_CACHE = {}


def packed_bwd_bytes(dtype):
    return int(_lib.load().nerfhip_mlp_packed_bwd_bytes(mlp_dtype_code(dtype)))


def pack_arg_tables(weights, biases):
    """Validate the 12 (weight, bias) tensors once and return ctypes pointer tables (wp, bp) for the pack entry points."""
    if len(weights) != 12 or len(biases) != 12:
        raise ValueError("need 12 weights and 12 biases")
    for i, (w, b) in enumerate(zip(weights, biases)):
        require_gpu(w, b)
        if tuple(w.shape) != PARAM_SHAPES[i] or tuple(b.shape) != (PARAM_SHAPES[i][0],):
            raise NerfHipError("fused MLP supports the reference default architecture only "
                               "(D=8, W=256, skips=[4], in 63/27); got %s for %s" % (tuple(w.shape), PARAM_ORDER[i]))
        if not (w.is_contiguous() and b.is_contiguous()):
            raise NerfHipError("NeRF parameters must be contiguous")
    wp = (ctypes.c_void_p * 12)(*[w.data_ptr() for w in weights])
    bp = (ctypes.c_void_p * 12)(*[b.data_ptr() for b in biases])
    return wp, bp


@device_guard
def pack_weights_raw(wp, bp, out, dtype):
    check(_lib.load().nerfhip_mlp_pack_weights(wp, bp, ptr(out), mlp_dtype_code(dtype), stream_ptr()), "nerfhip_mlp_pack_weights")


@device_guard
def pack_weights_bwd_raw(wp, bp, out, dtype):
    check(_lib.load().nerfhip_mlp_pack_weights_bwd(wp, bp, ptr(out), mlp_dtype_code(dtype), stream_ptr()),
          "nerfhip_mlp_pack_weights_bwd")


@device_guard
def pack_weights_train_raw(wp, bp, out, out_bwd, dtype):
    check(_lib.load().nerfhip_mlp_pack_weights_train(wp, bp, ptr(out), ptr(out_bwd), mlp_dtype_code(dtype), stream_ptr()),
          "nerfhip_mlp_pack_weights_train")


def alloc_acts(n_points, dtype, device):
    nbytes = int(_lib.load().nerfhip_mlp_act_bytes(int(n_points), mlp_dtype_code(dtype)))
    return torch.empty(nbytes, device=device, dtype=torch.uint8)


@device_guard
def mlp_fwd_embedded(x, packed, sigma_only, dtype, save=None):
    require_gpu(x)
    if x.dim() != 2 or x.stride(1) != 1:
        x = x.reshape(-1, x.shape[-1]).contiguous()
    n = x.shape[0]
    need = 63 if sigma_only else 90
    if x.shape[1] != need:
        raise ValueError("NeRF.forward expects %d input channels, got %d" % (need, x.shape[1]))
    out = torch.empty(n, 1 if sigma_only else 4, device=x.device, dtype=torch.float32)
    check(_lib.load().nerfhip_mlp_fwd_embedded(ptr(x), x.stride(0), n, ptr(packed), ptr(out), int(bool(sigma_only)),
                                               mlp_dtype_code(dtype), ptr(save), stream_ptr()),
          "nerfhip_mlp_fwd_embedded")
    return out


@device_guard
def mlp_fwd_rays(rays, z, packed, sigma_only, dtype, save=None):
    require_gpu(rays, z)
    rays, z = _c(rays), _c(z)
    B, S = z.shape
    out = torch.empty((B, S) if sigma_only else (B, S, 4), device=z.device, dtype=torch.float32)
    check(_lib.load().nerfhip_mlp_fwd_rays(ptr(rays), ptr(z), B, S, ptr(packed), ptr(out), int(bool(sigma_only)),
                                           mlp_dtype_code(dtype), ptr(save), stream_ptr()), "nerfhip_mlp_fwd_rays")
    return out


@device_guard
def mlp_fwd_rays_coarse(rays, n_samples, packed, sigma_only, dtype, use_disp=False, perturb=0.0, perturb_rand=None, save=None):
    """The coarse pass's sample_coarse_z + mlp_fwd_rays in one launch (nerfhip_mlp_fwd_rays_coarse): the depths are formed in the
    MLP kernel's prologue.  Returns (z (B,S), out)."""
    require_gpu(rays, perturb_rand)
    rays = _c(rays)
    B, S = rays.shape[0], int(n_samples)
    if perturb > 0:
        if perturb_rand is None:
            raise ValueError("perturb>0 needs perturb_rand")
        perturb_rand = _c(perturb_rand)
        if perturb_rand.numel() != B * S or perturb_rand.dtype != torch.float32:
            raise ValueError("mlp_fwd_rays_coarse: perturb_rand must hold B*S = %d fp32 draws, got %s %s"
                             % (B * S, tuple(perturb_rand.shape), perturb_rand.dtype))
    z = torch.empty(B, S, device=rays.device, dtype=torch.float32)
    out = torch.empty((B, S) if sigma_only else (B, S, 4), device=rays.device, dtype=torch.float32)
    check(_lib.load().nerfhip_mlp_fwd_rays_coarse(ptr(rays), ptr(perturb_rand if perturb > 0 else None), ptr(z), B, S,
                                                  int(bool(use_disp)), float(perturb), ptr(packed), ptr(out), int(bool(sigma_only)),
                                                  mlp_dtype_code(dtype), ptr(save), stream_ptr()), "nerfhip_mlp_fwd_rays_coarse")
    return z, out


# ------------------------------------------------------------------------------- render_rays in one launch
_render_fused = os.environ.get("NERFHIP_RENDER_FUSED", "1") != "0"


def set_render_fused(on):
    """Let render_rays / the fused training node use the single-launch render kernels where they apply (default) or keep them on
    the multi-launch path (A/B, bit-equality tests).  Returns the previous setting.  NERFHIP_RENDER_FUSED=0 in the environment."""
    global _render_fused
    prev, _render_fused = _render_fused, bool(on)
    return prev


def render_supported(B, S_c, N_i, dtype):
    """True when the single-launch render kernels take this shape (include/nerfhip.h nerfhip_render_supported: B % 4 == 0, 4 S_c and
    4 (S_c + N_i) multiples of the points per sub-pass) and set_render_fused has not switched them off."""
    if not _render_fused:
        return False
    return bool(_lib.load().nerfhip_render_supported(int(B), int(S_c), int(N_i), mlp_dtype_code(dtype)))


def _render_args(rays, S, N, packed_c, packed_f, use_disp, perturb, perturb_rand, noise_c, noise_f, noise_std, white_back, u, eps,
                 want_coarse, coarse_sigma_only=False):
    B = rays.shape[0]
    dev = rays.device
    f32 = dict(device=dev, dtype=torch.float32)
    a = _lib.RenderArgs()
    bufs = {"z_coarse": torch.empty(B, S, **f32), "opacity_coarse": torch.empty(B, **f32),
            "raw_coarse": torch.empty((B, S) if coarse_sigma_only else (B, S, 4), **f32)}
    if want_coarse:
        bufs.update(rgb_coarse=torch.empty(B, 3, **f32), depth_coarse=torch.empty(B, **f32))
    if N > 0:
        bufs.update(z_fine=torch.empty(B, S + N, **f32), raw_fine=torch.empty(B, S + N, 4, **f32), rgb_fine=torch.empty(B, 3, **f32),
                    depth_fine=torch.empty(B, **f32), opacity_fine=torch.empty(B, **f32))
    if perturb > 0:
        if perturb_rand is None:
            raise ValueError("perturb>0 needs perturb_rand")
        perturb_rand = _c(perturb_rand)
        _check_draw("render", "perturb_rand", perturb_rand, B * S)
    noise_c = None if noise_std == 0 else (_c(noise_c) if noise_c is not None else None)
    noise_f = None if (noise_std == 0 or N == 0) else (_c(noise_f) if noise_f is not None else None)
    if noise_std != 0 and (noise_c is None or (N > 0 and noise_f is None)):
        raise ValueError("render: noise_std != 0 needs the noise draws")
    _check_draw("render", "noise_coarse", noise_c, B * S)
    _check_draw("render", "noise_fine", noise_f, B * (S + N))
    u_stride = 0
    if u is not None and N > 0:
        u = _c(u)
        u_stride = N if u.dim() == 2 else 0
        _check_draw("render", "u", u, B * N if u.dim() == 2 else N)
    else:
        u = None
    a.rays, a.B, a.S_c, a.N_i = rays.data_ptr(), B, int(S), int(N)
    a.packed_coarse = packed_c.data_ptr()
    a.packed_fine = packed_f.data_ptr() if packed_f is not None else None
    for k, t in bufs.items():
        setattr(a, k, t.data_ptr())
    a.perturb_rand = perturb_rand.data_ptr() if perturb > 0 else None
    a.perturb, a.use_disp = float(perturb), int(bool(use_disp))
    a.noise_coarse = noise_c.data_ptr() if noise_c is not None else None
    a.noise_fine = noise_f.data_ptr() if noise_f is not None else None
    a.noise_std, a.white_back = float(noise_std), int(bool(white_back))
    a.u, a.u_stride, a.eps, a.row_total = (u.data_ptr() if u is not None else None), u_stride, float(eps), _row_total
    keep = (rays, packed_c, packed_f, perturb_rand, noise_c, noise_f, u)
    return a, bufs, keep


@device_guard
def render_fwd(rays, n_samples, n_importance, packed_coarse, packed_fine, dtype, use_disp=False, perturb=0.0, perturb_rand=None,
               noise_coarse=None, noise_fine=None, noise_std=0.0, white_back=False, u=None, eps=1e-5, want_coarse=True, test_time=False):
    """render_rays (rendering.py:58-244) for one ray chunk in ONE launch (nerfhip_render_fwd).  Returns the dict of every buffer
    the launch wrote: rgb / depth / opacity of both passes (rgb_coarse / depth_coarse only with want_coarse), z_* and raw_*.
    test_time (rendering.py:209-213, N_importance > 0): nerfhip_render_test_fwd — the coarse sub-passes run the network's sigma-only
    body, raw_coarse is (B, S) and the coarse pass leaves opacity_coarse only."""
    require_gpu(rays, perturb_rand, noise_coarse, noise_fine, u)
    rays = _c(rays)
    tt = bool(test_time) and int(n_importance) > 0
    a, bufs, keep = _render_args(rays, n_samples, n_importance, packed_coarse, packed_fine, use_disp, perturb, perturb_rand,
                                 noise_coarse, noise_fine, noise_std, white_back, u, eps, want_coarse and not tt, coarse_sigma_only=tt)
    if tt:
        check(_lib.load().nerfhip_render_test_fwd(ctypes.addressof(a), mlp_dtype_code(dtype), stream_ptr()), "nerfhip_render_test_fwd")
    else:
        check(_lib.load().nerfhip_render_fwd(ctypes.addressof(a), mlp_dtype_code(dtype), stream_ptr()), "nerfhip_render_fwd")
    return bufs


@device_guard
def render_train_fwd(rays, target, grad_scale, n_samples, n_importance, packed_coarse, packed_fine, dtype, acts_coarse, acts_fine,
                     use_disp=False, perturb=0.0, perturb_rand=None, noise_coarse=None, noise_fine=None, noise_std=0.0,
                     white_back=False, u=None, eps=1e-5, regen_enc=False):
    """The forward of a training step in ONE launch (nerfhip_render_train_fwd): render_fwd + saved activations + per pass the
    loss gradient and the compositing backward (g_raw_*) + out3 = [loss, psnr, mse].  Returns the dict of written buffers.
    regen_enc (bf16): the input-encoding slabs are not saved; the backward must be mlp_bwd_multi with (rays, z, S) entries."""
    require_gpu(rays, target, perturb_rand, noise_coarse, noise_fine, u)
    rays, target = _c(rays), _c(target)
    B, S, N = rays.shape[0], int(n_samples), int(n_importance)
    if target.numel() != 3 * B or target.dtype != torch.float32:
        raise ValueError("render_train_fwd: target must be (B,3) fp32")
    a, bufs, keep = _render_args(rays, S, N, packed_coarse, packed_fine, use_disp, perturb, perturb_rand, noise_coarse, noise_fine,
                                 noise_std, white_back, u, eps, True)
    dev = rays.device
    bufs["g_raw_coarse"] = torch.empty(B, S, 4, device=dev, dtype=torch.float32)
    if N > 0:
        bufs["g_raw_fine"] = torch.empty(B, S + N, 4, device=dev, dtype=torch.float32)
    bufs["out3"] = torch.empty(3, device=dev, dtype=torch.float32)
    a.g_raw_coarse = bufs["g_raw_coarse"].data_ptr()
    a.g_raw_fine = bufs["g_raw_fine"].data_ptr() if N > 0 else None
    a.out3 = bufs["out3"].data_ptr()
    a.save_coarse = acts_coarse.data_ptr()
    a.save_fine = acts_fine.data_ptr() if N > 0 else None
    a.target, a.grad_scale = target.data_ptr(), float(grad_scale)
    a.ticket = _ticket(dev).data_ptr()
    a.regen_enc = int(bool(regen_enc) and mlp_dtype_code(dtype) == BF16)
    check(_lib.load().nerfhip_render_train_fwd(ctypes.addressof(a), mlp_dtype_code(dtype), stream_ptr()), "nerfhip_render_train_fwd")
    return bufs


# ------------------------------------------------------------------------------- MLP backward (K2b)
@device_guard
def pack_weights_bwd(weights, biases, dtype, out=None):
    """W^T A-fragment stream for the backward chain + the fp32 fold block (12 weights, 12 biases, state_dict order)."""
    code = mlp_dtype_code(dtype)
    keep = [_c(w.detach()) for w in weights]
    keep_b = [_c(b.detach()) for b in biases]
    for w in keep + keep_b:
        require_gpu(w)
    if out is None:
        out = torch.empty(int(_lib.load().nerfhip_mlp_packed_bwd_bytes(code)), device=keep[0].device, dtype=torch.uint8)
    wp = (ctypes.c_void_p * 12)(*[k.data_ptr() for k in keep])
    bp = (ctypes.c_void_p * 12)(*[k.data_ptr() for k in keep_b])
    check(_lib.load().nerfhip_mlp_pack_weights_bwd(wp, bp, ptr(out), code, stream_ptr()), "nerfhip_mlp_pack_weights_bwd")
    return out


@device_guard
def mlp_bwd(g_out, out, packed_bwd, acts, dtype, shapes=PARAM_SHAPES, phases=7, workspace=None):
    """Gradients of all 24 parameter tensors given dL/d(out).  Returns ([gw0..gw11], [gb0..gb11], flat buffer).
    phases / workspace: measurement hooks (bench.py): run only some of the three kernels (bit 0 chain, 1 dW GEMM, 2 reduce)
    on caller-kept scratch buffers (dys, dw workspace, flat gradients)."""
    require_gpu(g_out, out)
    code = mlp_dtype_code(dtype)
    g_out = _c(g_out.float()).reshape(-1, 4)
    out = _c(out).reshape(-1, 4)
    n = out.shape[0]
    dev = out.device
    lib = _lib.load()
    if workspace is not None and "dys" in workspace:
        dys, ws = workspace["dys"], workspace["ws"]
    else:
        dys = torch.empty(int(lib.nerfhip_mlp_dy_bytes(n, code)), device=dev, dtype=torch.uint8)
        ws = torch.empty(int(lib.nerfhip_mlp_dw_workspace_bytes(n, code)), device=dev, dtype=torch.uint8)
        if workspace is not None:
            workspace["dys"], workspace["ws"] = dys, ws
    # one flat fp32 buffer for all 24 gradients (595,844 floats): autograd adopts the views as p.grad,
    # so a model's gradients are contiguous => ONE RCCL all-reduce per model, no flatten copies
    sizes = [s[0] * s[1] for s in shapes] + [s[0] for s in shapes]
    # n == 0: the C entry point launches nothing, so the gradients of an empty batch must be explicit zeros
    flat = (torch.zeros if n == 0 else torch.empty)(sum(sizes), device=dev, dtype=torch.float32)
    views, off = [], 0
    for sz in sizes:
        views.append(flat[off:off + sz])
        off += sz
    gw = [v.view(s) for v, s in zip(views[:12], shapes)]
    gb = views[12:]
    gwp = (ctypes.c_void_p * 12)(*[t.data_ptr() for t in gw])
    gbp = (ctypes.c_void_p * 12)(*[t.data_ptr() for t in gb])
    check(lib.nerfhip_mlp_bwd_phases(ptr(g_out), ptr(out), n, ptr(packed_bwd), ptr(acts), ptr(dys), ptr(ws), gwp, gbp, 0, code,
                                     int(phases), stream_ptr()), "nerfhip_mlp_bwd")
    return gw, gb, flat


@device_guard
def mlp_dx_embedded(dys, n, w_xyz1, w_xyz5, w_dir, dtype):
    """dL/dx (n, 90) of NeRF.forward on pre-embedded inputs from the dY slabs of the preceding mlp_bwd call."""
    code = mlp_dtype_code(dtype)
    if code == BF16_F8:
        raise NotImplementedError("nerfhip_mlp_dx_embedded reads bf16 / fp32 dY slabs ('bf16_f8' keeps dY only as e5m2: NeRF.forward "
                                  "saves in bf16 for calls whose input requires grad, models/mlp_autograd.py)")
    require_gpu(w_xyz1, w_xyz5, w_dir)
    gx = torch.empty(n, 90, device=dys.device, dtype=torch.float32)
    check(_lib.load().nerfhip_mlp_dx_embedded(ptr(dys), n, ptr(_c(w_xyz1.detach())), ptr(_c(w_xyz5.detach())), ptr(_c(w_dir.detach())),
                                              ptr(gx), 90, code, stream_ptr()), "nerfhip_mlp_dx_embedded")
    return gx


# ------------------------------------------------------------------------------- several models per launch (training step)
def pack_tables(models, dtype):
    """ctypes argument tables (weights, biases, forward images, W^T images) + the [(packed, packed_bwd)] buffers of `models` for
    the multi-model pack entry points (nerfhip_mlp_pack_weights_train_multi, nerfhip_train_prologue)."""
    if not 1 <= len(models) <= 4:
        raise ValueError("pack_models_train packs 1..4 models")
    tabs, bufs = [], []
    for m in models:
        if not m.is_default_arch():
            raise NotImplementedError("the fused HIP MLP implements the reference's default architecture only")
        wp, bp, dev = m._pack_args()
        tabs.append((wp, bp))
        bufs.append(m.train_buffers(dtype, dev))
    n = len(models)
    W = (ctypes.c_void_p * (12 * n))(*[t[0][i] for t in tabs for i in range(12)])
    Bv = (ctypes.c_void_p * (12 * n))(*[t[1][i] for t in tabs for i in range(12)])
    P = (ctypes.c_void_p * n)(*[b[0].data_ptr() for b in bufs])
    Pb = (ctypes.c_void_p * n)(*[b[1].data_ptr() for b in bufs])
    return W, Bv, P, Pb, bufs


def mark_packed(models):
    """the images just packed are those of the models' current weights (NeRF.check_pack_serial / train_step's freshness test)"""
    for m in models:
        m._packed_serial = getattr(m, "_weights_serial", 0)


def pack_models_train(models, dtype):
    """Forward + W^T images of every model in ONE launch (nerfhip_mlp_pack_weights_train_multi).  Returns [(packed, packed_bwd)]
    in the models' cached buffers (NeRF.packed_weights_train's)."""
    W, Bv, P, Pb, bufs = pack_tables(models, dtype)
    with torch.cuda.device(bufs[0][0].device):
        check(_lib.load().nerfhip_mlp_pack_weights_train_multi(W, Bv, P, Pb, len(models), mlp_dtype_code(dtype), stream_ptr()),
              "nerfhip_mlp_pack_weights_train_multi")
    mark_packed(models)
    return bufs


FLAT_GRAD_FLOATS = sum(s[0] * s[1] + s[0] for s in PARAM_SHAPES)      # 595,844: a model's 24 gradients (16-byte multiple)


def flat_grad_views(n_points, device, shapes=PARAM_SHAPES, out=None):
    """One flat fp32 buffer for a model's 24 gradients + the 12 weight and 12 bias views of it (the layout FlatAdam mirrors).
    out: the slice of a larger buffer to use as that flat buffer (mlp_bwd_multi: the models' buffers are consecutive slices of ONE
    allocation, so that the gradients of a whole step are one contiguous message for the all-reduce)."""
    sizes = [s[0] * s[1] for s in shapes] + [s[0] for s in shapes]
    if out is not None:
        flat = out
        if n_points == 0:
            flat.zero_()
    else:
        flat = (torch.zeros if n_points == 0 else torch.empty)(sum(sizes), device=device, dtype=torch.float32)
    views, off = [], 0
    for sz in sizes:
        views.append(flat[off:off + sz])
        off += sz
    return [v.view(s) for v, s in zip(views[:12], shapes)], views[12:], flat


def mlp_bwd_multi(entries, dtype, adam=None, phases=7, workspace=None, g_scale=None):
    """Backward of several models with ONE dW launch and ONE reduce launch (nerfhip_mlp_bwd_multi).
    entries: [(g_out (n,4), out (n,4), packed_bwd, acts)] or [(g_out, out, packed_bwd, acts, rays (B,8), z (B,S))] — the latter for
    a bf16 model whose forward did not save its input encodings (render_train_fwd(regen_enc=True)): the weight-gradient launch forms
    them again from the rays and the depths (nerfhip_mlp_bwd_multi_rays); n > 0.  adam: an _lib.AdamFused (the update then happens
    inside the reduce kernel).  g_scale: device scalar multiplying every g_out inside the chain kernels (the upstream gradient of the loss).
    Returns [(gw list, gb list, flat)] per model."""
    code = mlp_dtype_code(dtype)
    lib = _lib.load()
    M = len(entries)
    dev = entries[0][1].device
    gs, outs, ns, dys, grads = [], [], [], [], []
    enc, keep_enc = None, []
    for m, e in enumerate(entries):
        g_out, out = e[0], e[1]
        if len(e) > 4 and e[4] is not None:
            if enc is None:
                enc = _lib.EncSource()
            rays_e, z_e = _c(e[4]), _c(e[5])
            require_gpu(rays_e, z_e)
            if rays_e.dtype != torch.float32 or z_e.dtype != torch.float32 or z_e.dim() != 2 or rays_e.shape[0] != z_e.shape[0]:
                raise ValueError("mlp_bwd_multi: (rays (B,8), z (B,S)) fp32 expected")
            enc.rays[m], enc.z[m], enc.S[m] = rays_e.data_ptr(), z_e.data_ptr(), int(z_e.shape[1])
            keep_enc.append((rays_e, z_e))
        require_gpu(g_out, out)
        g_out = _c(g_out.float()).reshape(-1, 4)
        out = _c(out).reshape(-1, 4)
        gs.append(g_out)
        outs.append(out)
        ns.append(out.shape[0])
    with torch.cuda.device(dev):
        n_arr = (ctypes.c_int64 * M)(*ns)
        key = ("multi", tuple(ns), code)
        if workspace is not None and key in workspace:
            dys, ws = workspace[key]
        else:
            dys = [torch.empty(int(lib.nerfhip_mlp_dy_bytes(n, code)), device=dev, dtype=torch.uint8) for n in ns]
            ws = torch.empty(int(lib.nerfhip_mlp_dw_workspace_bytes_multi(n_arr, M, code)), device=dev, dtype=torch.uint8)
            if workspace is not None:
                workspace[key] = (dys, ws)
        # the M flat gradient buffers are consecutive slices of ONE allocation (`flat._base`): parallel.GradSync all-reduces a step's
        # gradients as one message
        joint = torch.empty(M * FLAT_GRAD_FLOATS, device=dev, dtype=torch.float32)
        for m, n in enumerate(ns):
            grads.append(flat_grad_views(n, dev, out=joint[m * FLAT_GRAD_FLOATS:(m + 1) * FLAT_GRAD_FLOATS]))
        vp = ctypes.c_void_p
        G = (vp * M)(*[t.data_ptr() for t in gs])
        O = (vp * M)(*[t.data_ptr() for t in outs])
        PB = (vp * M)(*[e[2].data_ptr() for e in entries])
        AC = (vp * M)(*[e[3].data_ptr() for e in entries])
        DY = (vp * M)(*[t.data_ptr() for t in dys])
        GW = (vp * (12 * M))(*[t.data_ptr() for g in grads for t in g[0]])
        GB = (vp * (12 * M))(*[t.data_ptr() for g in grads for t in g[1]])
        if adam is not None:
            for m in range(M):
                adam.grad_flat[m] = grads[m][2].data_ptr()
        if g_scale is not None:
            require_gpu(g_scale)
        check(lib.nerfhip_mlp_bwd_multi_rays(M, G, O, n_arr, PB, AC, DY, ptr(ws), GW, GB, 0, code, int(phases), ptr(g_scale),
                                             ctypes.addressof(adam) if adam is not None else None,
                                             ctypes.addressof(enc) if enc is not None else None, stream_ptr()), "nerfhip_mlp_bwd_multi_rays")
    return grads


# ------------------------------------------------------------------------------- layer-by-layer path (non-default NeRF shapes)
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def _rows(t):
    """(n, c) fp32 device tensor with unit column stride (a column slice of a wider tensor is fine: row stride = ld)"""
    if t.dim() != 2:
        raise ValueError("expected a (n, channels) tensor")
    if t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _off(t, col):
    """device pointer of t[:, col:] (row-major fp32)"""
    return ctypes.c_void_p(t.data_ptr() + 4 * col)


class _LinearAct(torch.autograd.Function):
    """act(cat(xs, -1) @ w.T + b) without the cat: one nerfhip_linear_fwd per input segment (the second accumulates on the first)."""

    @staticmethod
    def forward(ctx, act, dtype, w, b, *xs):
        xs = [_rows(x.float()) for x in xs]
        require_gpu(w, b, *xs)
        w, b = _c(w), _c(b)
        n, n_out, k_tot = xs[0].shape[0], w.shape[0], w.shape[1]
        if sum(x.shape[1] for x in xs) != k_tot or any(x.shape[0] != n for x in xs) or b.numel() != n_out:
            raise ValueError("linear: input segments %s do not match the weight %s" % ([tuple(x.shape) for x in xs], tuple(w.shape)))
        lib = _lib.load()
        code = mlp_dtype_code(dtype)
        y = torch.empty(n, n_out, device=w.device, dtype=torch.float32)
        col = 0
        for s, x in enumerate(xs):
            last = s == len(xs) - 1
            check(lib.nerfhip_linear_fwd(ptr(x), x.stride(0), _off(w, col), k_tot, ptr(b) if last else None, ptr(y), n_out, n,
                                         x.shape[1], n_out, act if last else ACT_NONE, int(s > 0), code, stream_ptr()),
                  "nerfhip_linear_fwd")
            col += x.shape[1]
        ctx.cfg = (act, code)
        ctx.save_for_backward(w, y, *xs)
        return y

    @staticmethod
    def backward(ctx, gy):
        w, y, *xs = ctx.saved_tensors
        act, code = ctx.cfg
        lib = _lib.load()
        gy = _c(gy.float())
        n, n_out, k_tot = y.shape[0], w.shape[0], w.shape[1]
        need_w, need_b = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        gw = gb = None
        gxs = []
        if (need_w or need_b) and n > 0:
            gw = torch.empty_like(w)
            gb = torch.empty(n_out, device=w.device, dtype=torch.float32)
            ws_bytes = max(lib.nerfhip_linear_bwd_weight_workspace_bytes(n, x.shape[1], n_out) for x in xs)
            ws = torch.empty(ws_bytes, device=w.device, dtype=torch.uint8)
        elif need_w or need_b:
            gw, gb = torch.zeros_like(w), torch.zeros(n_out, device=w.device, dtype=torch.float32)
        col = 0
        for s, x in enumerate(xs):
            if (need_w or need_b) and n > 0:
                check(lib.nerfhip_linear_bwd_weight(ptr(gy), n_out, ptr(y), n_out, act, ptr(x), x.stride(0), _off(gw, col), k_tot,
                                                    ptr(gb) if s == 0 else None, ptr(ws), n, x.shape[1], n_out, 0, code,
                                                    stream_ptr()), "nerfhip_linear_bwd_weight")
            gx = None
            if ctx.needs_input_grad[4 + s]:
                gx = torch.empty(n, x.shape[1], device=w.device, dtype=torch.float32)
                check(lib.nerfhip_linear_bwd_input(ptr(gy), n_out, ptr(y), n_out, act, _off(w, col), k_tot, ptr(gx), x.shape[1], n,
                                                   x.shape[1], n_out, 0, code, stream_ptr()), "nerfhip_linear_bwd_input")
            gxs.append(gx)
            col += x.shape[1]
        return (None, None, gw if need_w else None, gb if need_b else None) + tuple(gxs)


@device_guard
def linear_act(xs, weight, bias, act, dtype):
    """nn.Linear (+ activation) on the concatenation of the (n, c_i) tensors `xs` (nerf.py:59-81, 108-118), differentiable in
    weight, bias and every segment.  act: ACT_NONE | ACT_RELU | ACT_SIGMOID."""
    return _LinearAct.apply(int(act), dtype, weight, bias, *xs)
