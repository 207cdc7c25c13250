"""ctypes binding of libnerfhip.so (C ABI: include/nerfhip.h).

The product has NO CPU or eager-PyTorch fallback: if the HIP library is missing or a call is made
on a non-GPU tensor, this raises.  torch is imported first so that the library's
libamdhip64.so.7 dependency binds to the HIP runtime torch already loaded (one runtime per process).
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the dlopen below)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NERFHIP_LIB_PATH") or os.path.join(_PKG, "libnerfhip.so")   # override: A/B kernel builds

F32, BF16, BF16_F8 = 0, 1, 2

_c_void_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f32 = ctypes.c_float

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/nerfhip.h exactly
SIGNATURES = {
    "nerfhip_abi_version": [],
    "nerfhip_error_string": [_int],
    "nerfhip_posenc": [_c_void_p, _c_void_p, _i64, _int, _int, _c_void_p],
    "nerfhip_posenc_bwd": [_c_void_p, _c_void_p, _c_void_p, _i64, _int, _int, _c_void_p],
    "nerfhip_posenc_bands": [_c_void_p, _c_void_p, _c_void_p, _i64, _int, _int, _c_void_p],
    "nerfhip_posenc_bands_bwd": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _int, _int, _c_void_p],
    "nerfhip_sample_coarse_z": [_c_void_p, _c_void_p, _c_void_p, _i64, _int, _int, _f32, _c_void_p],
    "nerfhip_searchsorted_right": [_c_void_p, _c_void_p, _c_void_p, _i64, _int, _int, _c_void_p],
    "nerfhip_searchsorted_left": [_c_void_p, _c_void_p, _c_void_p, _i64, _int, _int, _c_void_p],
    "nerfhip_sample_pdf": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _int, _int, _f32,
                           _c_void_p],
    "nerfhip_sample_pdf_ex": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _int, _int, _f32,
                              _c_void_p, _c_void_p, _int, _c_void_p],
    "nerfhip_fine_z_ex": [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _int, _int, _f32,
                          _c_void_p, _c_void_p, _int, _c_void_p],
    "nerfhip_fine_z": [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _int, _int, _f32,
                       _c_void_p],
    "nerfhip_composite_fwd": [_c_void_p, _int, _c_void_p, _c_void_p, _c_void_p, _f32, _int, _c_void_p, _c_void_p,
                              _c_void_p, _c_void_p, _i64, _int, _c_void_p],
    "nerfhip_composite_bwd": [_c_void_p, _int, _c_void_p, _c_void_p, _c_void_p, _f32, _int, _c_void_p, _c_void_p,
                              _c_void_p, _c_void_p, _c_void_p, _i64, _int, _c_void_p],
    "nerfhip_mlp_packed_bytes": [_int],
    "nerfhip_mlp_pack_weights": [ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _c_void_p, _int, _c_void_p],
    "nerfhip_mlp_act_bytes": [_i64, _int],
    "nerfhip_mlp_fwd_embedded": [_c_void_p, _i64, _i64, _c_void_p, _c_void_p, _int, _int, _c_void_p, _c_void_p],
    "nerfhip_mlp_fwd_rays": [_c_void_p, _c_void_p, _i64, _int, _c_void_p, _c_void_p, _int, _int, _c_void_p, _c_void_p],
    "nerfhip_mlp_packed_bwd_bytes": [_int],
    "nerfhip_mlp_pack_weights_bwd": [ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _c_void_p, _int, _c_void_p],
    "nerfhip_mlp_pack_weights_train": [ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _c_void_p, _c_void_p, _int, _c_void_p],
    "nerfhip_mlp_dy_bytes": [_i64, _int],
    "nerfhip_mlp_dw_splits": [_i64, _int],
    "nerfhip_mlp_dw_workspace_bytes": [_i64, _int],
    "nerfhip_ray_directions": [_c_void_p, _int, _int, ctypes.c_double, _c_void_p],
    "nerfhip_get_rays": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p],
    "nerfhip_ndc_rays": [_int, _int, ctypes.c_double, _f32, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p],
    "nerfhip_gen_rays": [_c_void_p, _c_void_p, _i64, _i64, _int, _int, ctypes.c_double, _f32, _f32, _int, _f32, _c_void_p,
                         _c_void_p],
    "nerfhip_mse_psnr": [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "nerfhip_mlp_bwd_phases": [_c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                               ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _int, _int, _int, _c_void_p],
    "nerfhip_mlp_dx_embedded": [_c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _int, _c_void_p],
    "nerfhip_adam_step": [ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                          ctypes.POINTER(_c_void_p), ctypes.POINTER(_i64), _int, _c_void_p, _f32, _f32, _f32, _f32, _f32, _c_void_p],
    "nerfhip_mlp_bwd": [_c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                        ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _int, _int, _c_void_p],
    "nerfhip_sample_batch": [_c_void_p, _c_void_p, _c_void_p, _i64, _int, _int, ctypes.c_double, _f32, _f32, _int, _f32, _c_void_p,
                             _c_void_p, _c_void_p],
    "nerfhip_mlp_pack_weights_train_multi": [ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                             ctypes.POINTER(_c_void_p), _int, _int, _c_void_p],
    "nerfhip_render_supported": [_i64, _int, _int, _int],
    "nerfhip_render_test_fwd": [_c_void_p, _int, _c_void_p],
    "nerfhip_render_fwd": [_c_void_p, _int, _c_void_p],                  # (const nerfhip_render_args*: ctypes.addressof(RenderArgs))
    "nerfhip_render_train_fwd": [_c_void_p, _int, _c_void_p],
    "nerfhip_composite_train": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _f32, _int, _c_void_p, _f32, _c_void_p, _c_void_p,
                                _c_void_p, _c_void_p, _c_void_p, _i64, _int, _c_void_p],
    "nerfhip_composite_train_fine_z": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _f32, _int, _c_void_p, _f32, _c_void_p, _c_void_p,
                                       _c_void_p, _c_void_p, _c_void_p, _i64, _int, _c_void_p, _i64, _int, _f32, _c_void_p, _int, _c_void_p],
    "nerfhip_composite_train_loss": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _f32, _int, _c_void_p, _f32, _c_void_p, _c_void_p,
                                     _c_void_p, _c_void_p, _c_void_p, _i64, _int, _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "nerfhip_mlp_fwd_rays_coarse": [_c_void_p, _c_void_p, _c_void_p, _i64, _int, _int, _f32, _c_void_p, _c_void_p, _int, _int,
                                    _c_void_p, _c_void_p],
    "nerfhip_torch_draw_increment": [_i64, _int],
    "nerfhip_torch_draws": [_c_void_p, _int, _c_void_p, ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _int,
                            ctypes.POINTER(ctypes.c_uint64), _c_void_p],
    "nerfhip_train_prologue": [_c_void_p, _int, _c_void_p, ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _int,
                               ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                               ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _int, _int, _c_void_p],
    "nerfhip_mlp_dw_workspace_bytes_multi": [ctypes.POINTER(_i64), _int, _int],
    "nerfhip_mlp_dw_plan": [ctypes.POINTER(_i64), _int, _int, ctypes.POINTER(_int), ctypes.POINTER(_int)],
    "nerfhip_mlp_bwd_multi": [_int, ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), ctypes.POINTER(_i64),
                              ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _c_void_p,
                              ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _int, _int, _int, _c_void_p, _c_void_p, _c_void_p],
    "nerfhip_mlp_bwd_multi_rays": [_int, ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), ctypes.POINTER(_i64),
                                   ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _c_void_p,
                                   ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _int, _int, _int, _c_void_p, _c_void_p, _c_void_p,
                                   _c_void_p],
    "nerfhip_linear_fwd": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _i64, _int, _int, _int, _int, _int, _c_void_p],
    "nerfhip_linear_bwd_input": [_c_void_p, _i64, _c_void_p, _i64, _int, _c_void_p, _i64, _c_void_p, _i64, _i64, _int, _int, _int,
                                 _int, _c_void_p],
    "nerfhip_linear_bwd_weight_workspace_bytes": [_i64, _int, _int],
    "nerfhip_linear_bwd_weight": [_c_void_p, _i64, _c_void_p, _i64, _int, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p,
                                  _i64, _int, _int, _int, _int, _c_void_p],
}


class AdamFused(ctypes.Structure):
    """include/nerfhip.h: nerfhip_adam_fused"""
    _fields_ = [("n_models", _int), ("param", _c_void_p * 2), ("exp_avg", _c_void_p * 2), ("exp_avg_sq", _c_void_p * 2),
                ("grad_flat", _c_void_p * 2), ("state", _c_void_p), ("lr", _f32), ("beta1", _f32), ("beta2", _f32), ("eps", _f32),
                ("weight_decay", _f32)]

class Draw(ctypes.Structure):
    """include/nerfhip.h: nerfhip_draw"""
    _fields_ = [("kind", _int), ("numel", _i64), ("out", _c_void_p), ("range", ctypes.c_uint64)]


class RayBatch(ctypes.Structure):
    """include/nerfhip.h: nerfhip_ray_batch"""
    _fields_ = [("c2w", _c_void_p), ("rgbs_all", _c_void_p), ("rays", _c_void_p), ("rgbs", _c_void_p), ("H", _int), ("W", _int),
                ("focal", ctypes.c_double), ("near", _f32), ("far", _f32), ("use_ndc", _int), ("ndc_near_plane", _f32)]


class RenderArgs(ctypes.Structure):
    """include/nerfhip.h: nerfhip_render_args"""
    _fields_ = [("rays", _c_void_p), ("B", _i64), ("S_c", _int), ("N_i", _int), ("packed_coarse", _c_void_p), ("packed_fine", _c_void_p),
                ("z_coarse", _c_void_p), ("raw_coarse", _c_void_p), ("z_fine", _c_void_p), ("raw_fine", _c_void_p),
                ("save_coarse", _c_void_p), ("save_fine", _c_void_p), ("perturb_rand", _c_void_p), ("perturb", _f32), ("use_disp", _int),
                ("noise_coarse", _c_void_p), ("noise_fine", _c_void_p), ("noise_std", _f32), ("white_back", _int), ("u", _c_void_p),
                ("u_stride", _i64), ("eps", _f32), ("row_total", _int), ("rgb_coarse", _c_void_p), ("depth_coarse", _c_void_p),
                ("opacity_coarse", _c_void_p), ("rgb_fine", _c_void_p), ("depth_fine", _c_void_p), ("opacity_fine", _c_void_p),
                ("target", _c_void_p), ("grad_scale", _f32), ("g_raw_coarse", _c_void_p), ("g_raw_fine", _c_void_p), ("out3", _c_void_p),
                ("ticket", _c_void_p), ("regen_enc", _int)]


class EncSource(ctypes.Structure):
    """include/nerfhip.h: nerfhip_enc_source"""
    _fields_ = [("rays", _c_void_p * 2), ("z", _c_void_p * 2), ("S", _int * 2)]


DRAW_UNIFORM, DRAW_NORMAL, DRAW_RANDINT = 0, 1, 2

_RESTYPES = {"nerfhip_error_string": ctypes.c_char_p, "nerfhip_torch_draw_increment": ctypes.c_uint64, "nerfhip_mlp_packed_bytes": ctypes.c_size_t,
             "nerfhip_mlp_act_bytes": ctypes.c_size_t, "nerfhip_mlp_packed_bwd_bytes": ctypes.c_size_t,
             "nerfhip_mlp_dy_bytes": ctypes.c_size_t, "nerfhip_mlp_dw_workspace_bytes": ctypes.c_size_t,
             "nerfhip_mlp_dw_workspace_bytes_multi": ctypes.c_size_t,
             "nerfhip_linear_bwd_weight_workspace_bytes": ctypes.c_size_t}

_lib = None


class NerfHipError(RuntimeError):
    pass


def load():
    """dlopen libnerfhip.so and bind every symbol of the header.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NerfHipError(
            "libnerfhip.so not built: run `python -m nerf_pl_amd.build` (needs hipcc, gfx950). "
            "There is no CPU/eager fallback for the hot path.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    if lib.nerfhip_abi_version() != 3:
        raise NerfHipError("libnerfhip ABI version mismatch")
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().nerfhip_error_string(code)
        raise NerfHipError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", code))


def stream_ptr():
    """Raw hipStream_t of torch's current stream on the CURRENT device (kernels launch there: DDP overlap, graph
    capture).  Every operator runs under `device_guard`, which makes the tensors' device the current one first."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _first_cuda_tensor(args):
    for a in args:
        if torch.is_tensor(a) and a.is_cuda:
            return a
        if isinstance(a, (list, tuple)):
            t = _first_cuda_tensor(a)
            if t is not None:
                return t
    return None


def device_guard(fn):
    """Run `fn` with the device of its first CUDA tensor argument as the current device (what ATen's ops do
    implicitly): the HIP launch and `stream_ptr()` then refer to the GPU that owns the buffers, also when the
    caller's current device is a different one."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kw):
        t = _first_cuda_tensor(args)
        if t is None:
            t = _first_cuda_tensor(tuple(kw.values()))
        if t is None or t.device.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(t.device):
            return fn(*args, **kw)
    return wrapped


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_gpu(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NerfHipError("nerf_pl_amd runs on MI355X only: got a %s tensor (no CPU fallback)" % t.device)
        if t.dtype != torch.float32:
            raise NerfHipError("expected float32 tensor, got %s" % t.dtype)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise NerfHipError("tensors of one call live on different GPUs (%s vs %s)" % (dev, t.device))
    if dev is not None and dev.index != torch.cuda.current_device():
        raise NerfHipError("launch on %s while the current device is cuda:%d (operator missing its device_guard)"
                           % (dev, torch.cuda.current_device()))
