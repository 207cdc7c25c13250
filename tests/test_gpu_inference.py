"""GPU: whole-image inference (eval.py:58-86 mirror) — batched_inference and the hipGraph-replayed
GraphRenderer against the CPU oracle's test_time render, incl. ragged tail chunks, and full-size
(32768-ray chunk) size-independent properties."""
import pytest
import torch

from oracle import nerf_oracle as O
from tests.helpers import build_models

pytestmark = pytest.mark.gpu


def _oracle_test_time(params, rays, S, N, white_back):
    return O.render_rays(params, rays, S, False, 0, 0, N, white_back, True)


@pytest.mark.parametrize("B", [1, 100, 257])
def test_batched_inference_matches_oracle_fp32(dev, B):
    from nerf_pl_amd.inference import batched_inference
    params = [O.make_params(11, 6.0, 0.3), O.make_params(12, 6.0, 0.3)]
    rays = O.make_rays(5, B, "blender")
    ref = _oracle_test_time(params, rays, 64, 64, True)
    ms, emb = build_models(params, dev, "fp32")
    res = batched_inference(ms, emb, rays.to(dev), 64, 64, False, 1024 * 32, True)
    assert set(res.keys()) == set(ref.keys()) == {"opacity_coarse", "rgb_fine", "depth_fine", "opacity_fine"}
    for k, v in ref.items():
        assert res[k].shape == v.shape
        assert torch.allclose(res[k].cpu(), v, rtol=1e-4, atol=1e-4), k


def test_graph_renderer_equals_eager_and_handles_tail(dev):
    """hipGraph replay == eager launches bit for bit (same kernels, same inputs); the tail chunk is padded."""
    from nerf_pl_amd.inference import GraphRenderer, batched_inference
    params = [O.make_params(21, 6.0, 0.3), O.make_params(22, 6.0, 0.3)]
    ms, emb = build_models(params, dev, "bf16")
    chunk = 512
    gr = GraphRenderer(ms, emb, 64, 128, False, True, chunk=chunk)
    rays = O.make_rays(9, 2 * chunk + 77, "blender").to(dev)            # two full chunks + ragged tail
    got = gr(rays)
    want = batched_inference(ms, emb, rays, 64, 128, False, 1024 * 32, True)
    for k in want:
        assert got[k].shape == want[k].shape
        assert torch.equal(got[k], want[k]), k
    # weights change => the graph must see the re-packed stream
    with torch.no_grad():
        for p in ms[1].parameters():
            p.mul_(0.5)
    got2 = gr(rays[:chunk])
    want2 = batched_inference(ms, emb, rays[:chunk], 64, 128, False, 1024 * 32, True)
    assert torch.equal(got2["rgb_fine"], want2["rgb_fine"])
    assert not torch.equal(got2["rgb_fine"], got["rgb_fine"][:chunk])


def test_render_to_host_overlapped_equals_device_path(dev, tmp_path):
    """N3: the D2H-overlapped whole-image render (pinned host buffers filled chunk by chunk on a copy stream) returns the
    same pixels as the device path, incl. the ragged tail chunk; and the eval loop's writers consume it."""
    import numpy as np
    from nerf_pl_amd import imageio_min as io
    from nerf_pl_amd.inference import GraphRenderer, save_image_outputs
    params = [O.make_params(21, 6.0, 0.3), O.make_params(22, 6.0, 0.3)]
    ms, emb = build_models(params, dev, "bf16")
    h, w, chunk = 40, 51, 512                                  # 2040 rays = 3 full chunks + a tail of 504
    gr = GraphRenderer(ms, emb, 64, 128, False, True, chunk=chunk)
    rays = O.make_rays(9, h * w, "blender").to(dev)
    want = gr(rays)
    for _ in range(2):                                         # twice: the copy stream / pinned buffers are reused
        host = gr.render_to_host(rays, keys=("rgb_fine", "depth_fine", "opacity_fine"))
        for k, v in host.items():
            assert v.is_pinned() and not v.is_cuda
            assert torch.equal(v, want[k].cpu()), k
    img8 = save_image_outputs(host, h, w, str(tmp_path), 0, save_depth=True)
    assert np.array_equal(io.read_png(str(tmp_path / "000.png")), img8)
    d, _ = io.read_pfm(str(tmp_path / "depth_000.pfm"))
    assert np.array_equal(d, want["depth_fine"].reshape(h, w).cpu().numpy())


def test_full_chunk_properties_bf16(dev):
    """BASELINE-size chunk (32768 rays x (64+128)): size-independent properties instead of an oracle run:
    per-ray independence (any sub-batch renders identically), opacity in [0,1], rgb in [0,1] (white_back
    adds 1-opacity), depth within [0, far]."""
    from nerf_pl_amd.inference import batched_inference
    params = [O.make_params(31, 6.0, 0.3), O.make_params(32, 6.0, 0.3)]
    ms, emb = build_models(params, dev, "bf16")
    rays = O.make_rays(1, 32768, "blender").to(dev)
    full = batched_inference(ms, emb, rays, 64, 128, False, 1024 * 32, True)
    sub = batched_inference(ms, emb, rays[1000:1300], 64, 128, False, 1024 * 32, True)
    for k in full:
        assert torch.equal(full[k][1000:1300], sub[k]), k
    assert torch.isfinite(full["rgb_fine"]).all()
    assert (full["opacity_fine"] >= 0).all() and (full["opacity_fine"] <= 1 + 1e-5).all()
    assert (full["rgb_fine"] >= -1e-5).all() and (full["rgb_fine"] <= 1 + 1e-4).all()
    assert (full["depth_fine"] >= 0).all() and (full["depth_fine"] <= 6.0 + 1e-3).all()


def test_empty_and_single_ray_batches(dev):
    """Edge sizes through the whole path: B = 0 (empty chunk after sharding) and B = 1."""
    from nerf_pl_amd.models import render_rays
    params = [O.make_params(41, 6.0, 0.3), O.make_params(42, 6.0, 0.3)]
    ms, emb = build_models(params, dev, "fp32")
    with torch.no_grad():
        out = render_rays(ms, emb, torch.zeros(0, 8, device=dev), 64, False, 0, 0, 64, 32768, True, test_time=True)
    assert out["rgb_fine"].shape == (0, 3) and out["opacity_coarse"].shape == (0,)
    rays = O.make_rays(2, 1, "blender")
    ref = O.render_rays(params, rays, 64, False, 0, 0, 64, True, False)
    with torch.no_grad():
        got = render_rays(ms, emb, rays.to(dev), 64, False, 0, 0, 64, 32768, True)
    for k, v in ref.items():
        assert torch.allclose(got[k].cpu(), v, rtol=1e-4, atol=1e-4), k


def test_sigma_grid_matches_reference_lattice_order(dev):
    """N4: sigma_grid == the reference's dense query (extract_color_mesh.py:113-140): same lattice (np.meshgrid 'xy'
    order => [iy, ix, iz]), same values as the oracle MLP on the embedded lattice points, clamped at 0."""
    import numpy as np
    from nerf_pl_amd.grid import sigma_grid
    p = O.make_params(51, 6.0, 0.3)
    (m,), _ = build_models([p], dev, "fp32")
    N = 9
    xr, yr, zr = (-1.2, 1.2), (-1.0, 1.3), (-0.7, 1.1)
    got = sigma_grid(m, N, xr, yr, zr).cpu()
    x, y, z = (np.linspace(lo, hi, N) for lo, hi in (xr, yr, zr))
    xyz = torch.FloatTensor(np.stack(np.meshgrid(x, y, z), -1).reshape(-1, 3))
    ref = O.mlp_forward(p, O.posenc(xyz, 10), sigma_only=True)[:, 0].clamp(min=0).reshape(N, N, N)
    assert got.shape == (N, N, N)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5), (got - ref).abs().max().item()
    # chunked launches == single launch (ragged last chunk), bf16 finite and close
    assert torch.equal(sigma_grid(m, N, xr, yr, zr, rows_per_launch=7).cpu(), got)
    m.mlp_dtype = "bf16"
    gb = sigma_grid(m, N, xr, yr, zr, clamp=False).cpu()
    raw = O.mlp_forward(p, O.posenc(xyz, 10), sigma_only=True)[:, 0].reshape(N, N, N)
    assert (gb - raw).abs().max().item() <= 3e-2 * max(1.0, raw.abs().max().item())
