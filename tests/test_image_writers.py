"""Host-side writers of the eval loop (N3): PFM byte-identical to the reference's datasets/depth_utils.py `save_pfm`
(executed from /root/reference when present, and against a committed fixture everywhere), PNG round trip, depth bytes."""
import hashlib
import importlib.util
import os

import numpy as np
import pytest

from nerf_pl_amd import imageio_min as io

REF_DU = "/root/reference/datasets/depth_utils.py"


def _depth(h=37, w=53, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.random((h, w), dtype=np.float32) * 6.0).astype(np.float32)


def test_pfm_roundtrip_and_fixture(tmp_path):
    d = _depth()
    p = str(tmp_path / "d.pfm")
    io.save_pfm(p, d)
    raw = open(p, "rb").read()
    assert raw.startswith(b"Pf\n53 37\n-1.000000\n") and len(raw) == len(b"Pf\n53 37\n-1.000000\n") + 37 * 53 * 4
    # rows are stored bottom-to-top
    assert np.frombuffer(raw[-53 * 4:], dtype="<f4").tolist() == d[0].tolist()
    back, scale = io.read_pfm(p)
    assert scale == 1.0 and back.dtype == np.float32 and np.array_equal(back, d)
    # fixture: sha256 of the file the REFERENCE's save_pfm (datasets/depth_utils.py:43-69) writes for this array, minted in
    # the build container (test_pfm_bytes_identical_to_reference re-checks it there against the live reference)
    assert hashlib.sha256(raw).hexdigest() == "5426d477e73895a38eaf24cadd04f5c356eb02c4eea34c4f99556d11ef3c5969"
    col = np.stack([d, d * 0.5, d * 0.25], -1)
    io.save_pfm(p, col, scale=2)
    back, scale = io.read_pfm(p)
    assert scale == 2.0 and np.array_equal(back, col)
    with pytest.raises(Exception):
        io.save_pfm(p, d.astype(np.float64))


@pytest.mark.skipif(not os.path.exists(REF_DU), reason="reference tree not present")
def test_pfm_bytes_identical_to_reference(tmp_path):
    spec = importlib.util.spec_from_file_location("ref_depth_utils", REF_DU)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for arr in (_depth(), _depth(8, 8, 3), np.stack([_depth(5, 7, 1)] * 3, -1), _depth(4, 6, 2)[..., None]):
        a, b = str(tmp_path / "a.pfm"), str(tmp_path / "b.pfm")
        ref.save_pfm(a, arr)
        io.save_pfm(b, arr)
        assert open(a, "rb").read() == open(b, "rb").read()
        ra, sa = ref.read_pfm(a)
        rb, sb = io.read_pfm(b)
        assert sa == sb and np.array_equal(ra, rb)


def test_png_roundtrip(tmp_path):
    rng = np.random.default_rng(1)
    for shape in ((31, 45, 3), (16, 16), (5, 9, 4)):
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        p = str(tmp_path / "x.png")
        io.write_png(p, img)
        assert np.array_equal(io.read_png(p), img)
        try:                                                 # an independent decoder, when the image has one
            from PIL import Image
            assert np.array_equal(np.asarray(Image.open(p)), img)
        except ImportError:
            pass
    with pytest.raises(ValueError):
        io.write_png(str(tmp_path / "y.png"), np.zeros((4, 4, 3), dtype=np.float32))


def test_depth_bytes_and_save_image_outputs(tmp_path):
    import torch
    from nerf_pl_amd.inference import save_image_outputs
    h, w = 12, 20
    rgb = torch.rand(h * w, 3)
    depth = torch.rand(h * w) * 6
    depth[3] = float("nan")
    img8 = save_image_outputs({"rgb_fine": rgb, "depth_fine": depth}, h, w, str(tmp_path), 7, save_depth=True)
    assert np.array_equal(io.read_png(str(tmp_path / "007.png")), img8)
    assert np.array_equal(img8, (rgb.reshape(h, w, 3).numpy() * 255).astype(np.uint8))      # eval.py:139
    d, _ = io.read_pfm(str(tmp_path / "depth_007.pfm"))
    assert np.array_equal(d, np.nan_to_num(depth.reshape(h, w).numpy()))                    # eval.py:133
    save_image_outputs({"rgb_fine": rgb, "depth_fine": depth}, h, w, str(tmp_path), 8, save_depth=True, depth_format="bytes")
    assert open(str(tmp_path / "depth_008"), "rb").read() == np.nan_to_num(depth.reshape(h, w).numpy()).tobytes()
