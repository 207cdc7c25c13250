"""CPU: libnerfhip.so builds/loads without a GPU and exports every symbol include/nerfhip.h declares
(no compute calls here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "nerfhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nerfhip_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from nerf_pl_amd import build
    build.build(verbose=False)
    from nerf_pl_amd import _lib
    return _lib.load()


def test_header_declares_expected_surface():
    syms = _header_symbols()
    for must in ("nerfhip_posenc", "nerfhip_searchsorted_right", "nerfhip_sample_pdf", "nerfhip_composite_fwd",
                 "nerfhip_composite_bwd", "nerfhip_mlp_fwd_rays", "nerfhip_mlp_fwd_embedded",
                 "nerfhip_mlp_pack_weights"):
        assert must in syms


def test_library_exports_every_header_symbol(lib):
    from nerf_pl_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (nerfhip_[a-z0-9_]+)", out))
    missing = [s for s in _header_symbols() if s not in exported]
    assert not missing, missing
    # and the ctypes table binds exactly the header's surface
    assert sorted(_lib.SIGNATURES) == _header_symbols()


def test_abi_basics(lib):
    assert lib.nerfhip_abi_version() == 3
    assert lib.nerfhip_error_string(0) == b"success"
    assert b"aligned" in lib.nerfhip_error_string(-3)
    # packed stream sizes: multiples of the 32 KiB ring chunk (mlp_layout.h)
    for code in (0, 1):
        n = lib.nerfhip_mlp_packed_bytes(code)
        assert n > 0 and n % 32768 == 0
    assert lib.nerfhip_mlp_packed_bytes(7) == 0
    # 1,056 (fp32: 2,112) fragment pieces of the 11 kernel layers (xyz_encoding_final is folded into the dir layer) + the
    # chunk-padded bias block between layers 4 and 5 (mlp_layout.h)
    assert lib.nerfhip_mlp_packed_bytes(1) == lib.nerfhip_mlp_packed_bytes(2) == (1056 + 32) * 1024
    assert lib.nerfhip_mlp_packed_bytes(0) == (2112 + 64) * 1024
    # W^T stream of the chain (9 layers, chunk-padded) + the 385-piece fp32 fold block
    assert lib.nerfhip_mlp_packed_bwd_bytes(1) == (992 + 385) * 1024 and lib.nerfhip_mlp_packed_bwd_bytes(0) == (1952 + 385) * 1024


def test_no_cpu_fallback():
    """The product must refuse CPU tensors instead of silently computing somewhere else."""
    import torch
    from nerf_pl_amd import ops
    from nerf_pl_amd._lib import NerfHipError
    with pytest.raises(NerfHipError):
        ops.posenc(torch.zeros(4, 3), 10)
    from nerf_pl_amd.models import Embedding, NeRF, render_rays
    with pytest.raises(NerfHipError):
        render_rays([NeRF()], [Embedding(3, 10), Embedding(3, 4)], torch.zeros(4, 8), 8)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nerf_pl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dp, f)


def test_install_registers_reference_module_names():
    """nerf_pl_amd.install() makes the reference's imports (train.py:10-11, eval.py:9-10, rendering.py:2) resolve here."""
    import importlib
    import sys
    saved = {k: sys.modules.get(k) for k in ("models", "models.nerf", "models.rendering", "torchsearchsorted")}
    try:
        import nerf_pl_amd
        nerf_pl_amd.install()
        nerf = importlib.import_module("models.nerf")
        rendering = importlib.import_module("models.rendering")
        tss = importlib.import_module("torchsearchsorted")
        from nerf_pl_amd.models import nerf as ours_nerf, rendering as ours_r
        assert nerf.Embedding is ours_nerf.Embedding and nerf.NeRF is ours_nerf.NeRF
        assert rendering.render_rays is ours_r.render_rays and rendering.__all__ == ['render_rays']
        assert callable(tss.searchsorted)
        # constructor surface and state_dict keys of the reference (nerf.py:42-81)
        m = nerf.NeRF(D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4])
        keys = list(m.state_dict().keys())
        assert keys[:2] == ["xyz_encoding_1.0.weight", "xyz_encoding_1.0.bias"] and "rgb.0.bias" in keys and len(keys) == 24
        e = nerf.Embedding(3, 10)
        assert e.out_channels == 63 and len(e.freq_bands) == 10 and e.funcs[0] is __import__("torch").sin
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_header_is_plain_c_and_links(lib, tmp_path):
    """include/nerfhip.h is the boundary a non-Python host binds: it must compile as strict C99 (and as C++), and a C program
    referencing every declared entry point must link against libnerfhip.so (no GPU needed: nothing is called)."""
    from nerf_pl_amd import _lib
    src = tmp_path / "use_all.c"
    body = "\n".join("    p[%d] = (void (*)(void))%s;" % (i, s) for i, s in enumerate(_header_symbols()))
    src.write_text('#include "nerfhip.h"\nint main(void) {\n    void (*p[%d])(void);\n%s\n    return p[0] == 0;\n}\n'
                   % (len(_header_symbols()), body))
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-Wno-pedantic", "-I", inc, "-c", str(src),
                    "-o", str(tmp_path / "use_all.o")], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-I", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "use_all_cc.o")],
                   check=True)
    libdir = os.path.dirname(_lib.LIB_PATH)
    hip = "/opt/rocm/lib"
    subprocess.run(["gcc", str(tmp_path / "use_all.o"), "-o", str(tmp_path / "use_all"), "-L", libdir, "-lnerfhip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath," + hip, "-L", hip, "-lamdhip64"], check=True)


def test_linear_entry_points_validate_arguments_without_a_gpu(lib):
    """The layer-by-layer path's C entry points (non-default NeRF shapes): argument validation returns NERFHIP_E_BADARG before
    anything is launched, and the weight-gradient workspace follows the documented split plan (whole 128 x 128/256 tiles per
    split + one bias row per split)."""
    null = None
    # n_in < 1, ld < n_in, unknown activation, unknown dtype: refused (-1) with null pointers never dereferenced
    assert lib.nerfhip_linear_fwd(null, 8, null, 8, null, null, 8, 16, 0, 8, 0, 0, 0, null) == -1
    assert lib.nerfhip_linear_fwd(null, 4, null, 8, null, null, 8, 16, 8, 8, 0, 0, 0, null) == -1
    assert lib.nerfhip_linear_fwd(null, 8, null, 8, null, null, 8, 16, 8, 8, 3, 0, 0, null) == -1
    assert lib.nerfhip_linear_fwd(null, 8, null, 8, null, null, 8, 16, 8, 8, 0, 0, 9, null) == -1
    assert lib.nerfhip_linear_fwd(null, 8, null, 8, null, null, 8, 16, 8, 8, 0, 0, 0, null) == -1      # null tensors, n > 0
    assert lib.nerfhip_linear_fwd(null, 8, null, 8, null, null, 8, 0, 8, 8, 0, 0, 0, null) == 0        # n == 0: nothing to do
    assert lib.nerfhip_linear_bwd_input(null, 8, null, 8, 1, null, 8, null, 8, 16, 8, 8, 0, 0, null) == -1   # ReLU needs y
    assert lib.nerfhip_linear_bwd_weight(null, 8, null, 8, 0, null, 8, null, 8, null, null, 16, 8, 8, 0, 0, null) == -1
    ws = lib.nerfhip_linear_bwd_weight_workspace_bytes
    assert ws(0, 8, 8) == 0 and ws(16, 0, 8) == 0
    one = ws(100, 63, 256)            # 1 split: 256 x 128 tile floats + 256 bias floats
    assert one == (256 * 128 + 256) * 4
    big = ws(196608, 256, 256)        # 2 x 1 tiles of 128 x 256: 512 two-tile workgroup equivalents => 256 splits of 768 points
    assert big == 256 * (256 * 256 + 256) * 4
    assert ws(196608, 319, 256) == 128 * (256 * 512 + 256) * 4      # two 256-column tiles per row of tiles: half the splits


def test_bwd_multi_rays_refuses_what_it_cannot_regenerate(lib):
    """nerfhip_mlp_bwd_multi_rays validates its encoding source before anything is launched or dereferenced (host logic: runs
    without a GPU): only the bf16 arithmetic regenerates, S must be a whole number of 32-point tiles, n a whole number of
    256-point workgroups and of rays, the arrays 16-byte aligned."""
    import ctypes
    from nerf_pl_amd import _lib
    vp = ctypes.c_void_p
    fake = 0x10000                                                     # never dereferenced: every call below returns from the checks
    one = (vp * 1)(fake)
    grads = (vp * 12)(*([fake] * 12))
    F32, BF16, BF16_F8 = 0, 1, 2

    def call(n, dtype, rays=fake, z=fake, S=64):
        enc = _lib.EncSource()
        enc.rays[0], enc.z[0], enc.S[0] = rays, z, S
        n_arr = (ctypes.c_int64 * 1)(n)
        return lib.nerfhip_mlp_bwd_multi_rays(1, one, one, n_arr, one, one, one, vp(fake), grads, grads, 0, dtype, 7, None, None,
                                              ctypes.addressof(enc), None)
    assert call(64 * 64, F32) == -2 and call(64 * 64, BF16_F8) == -2    # NERFHIP_E_UNSUPPORTED
    assert call(64 * 48, BF16, S=48) == -1                             # S not a multiple of 32
    assert call(64 * 64 + 32, BF16) == -1                              # n not a multiple of 256
    assert call(64 * 64, BF16, z=fake + 4) == -3                       # NERFHIP_E_ALIGN
    assert call(64 * 64, BF16, z=None) == -1                           # rays without depths
    assert call(64 * 64, 7) == -2
