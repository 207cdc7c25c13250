"""GPU: render_rays in ONE launch (csrc/mlp_render_kernel.h: nerfhip_render_fwd / nerfhip_render_train_fwd) against the launches it
replaces — nerfhip_mlp_fwd_rays_coarse -> nerfhip_composite_fwd -> nerfhip_fine_z -> nerfhip_mlp_fwd_rays -> nerfhip_composite_fwd, and
for the training forward nerfhip_composite_train_fine_z / nerfhip_composite_train_loss — BIT FOR BIT: every per-point intermediate
(depths, rgb sigma), every rendered output, the saved activations, d loss / d raw, loss / PSNR, and through them every parameter
gradient.  The multi-launch path is itself pinned to the reference-minted vectors (tests/test_gpu_parity.py, test_gpu_training.py)."""
import ctypes

import pytest
import torch

from oracle import nerf_oracle as O
from tests.helpers import build_models

pytestmark = pytest.mark.gpu


def _models(dev, dtype, seeds=(5, 6)):
    return build_models([O.make_params(seeds[0], 4.0, 0.2), O.make_params(seeds[1], 4.0, 0.2)], dev, dtype)


def _draws(B, S, N, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {"perturb_rand": torch.rand(B, S, generator=g).to(dev), "noise_coarse": torch.randn(B, S, generator=g).to(dev),
            "u": torch.rand(B, max(N, 1), generator=g)[:, :N].contiguous().to(dev), "noise_fine": torch.randn(B, S + N, generator=g).to(dev)}


CASES = [  # dtype, B, S, N, perturb, use_disp, noise_std, white_back, random u, kind
    ("bf16", 1024, 64, 128, 1.0, False, 0.0, True, True, "blender"),
    ("bf16", 64, 64, 64, 0.0, False, 1.0, False, False, "ndc"),
    ("bf16", 8, 64, 0, 1.0, True, 1.0, True, False, "blender"),
    ("bf16", 12, 128, 64, 0.5, False, 0.0, False, True, "blender"),
    ("bf16", 4, 64, 192, 1.0, False, 0.0, True, True, "blender"),
    ("fp32", 256, 64, 128, 1.0, False, 0.0, True, True, "blender"),
    ("fp32", 4, 32, 32, 0.0, True, 1.0, False, False, "blender"),
    ("fp32", 20, 64, 0, 1.0, False, 1.0, True, False, "ndc"),
    ("fp32", 8, 96, 32, 1.0, False, 0.0, True, True, "blender"),
]


@pytest.mark.parametrize("dtype,B,S,N,perturb,use_disp,noise_std,white_back,rand_u,kind", CASES)
def test_render_fwd_is_the_launches_it_replaces(dev, dtype, B, S, N, perturb, use_disp, noise_std, white_back, rand_u, kind):
    from nerf_pl_amd import ops
    ms, _ = _models(dev, dtype)
    rays = O.make_rays(3, B, kind).to(dev)
    d = _draws(B, S, N, dev)
    pr = d["perturb_rand"] if perturb > 0 else None
    u = d["u"] if (rand_u and N > 0) else None
    assert ops.render_supported(B, S, N, dtype)
    pc, pf = ms[0].packed_weights(dtype), ms[1].packed_weights(dtype)
    # ---- the launches ----
    z, raw_c = ops.mlp_fwd_rays_coarse(rays, S, pc, False, dtype, use_disp, perturb, pr)
    w_c, opac_c, rgb_c, depth_c = ops.composite(raw_c, z, rays, d["noise_coarse"], noise_std, white_back)
    want = {"z_coarse": z, "raw_coarse": raw_c, "rgb_coarse": rgb_c, "depth_coarse": depth_c, "opacity_coarse": opac_c}
    if N > 0:
        zf = ops.fine_z(z, w_c, N, u=u)
        raw_f = ops.mlp_fwd_rays(rays, zf, pf, False, dtype)
        _, opac_f, rgb_f, depth_f = ops.composite(raw_f, zf, rays, d["noise_fine"], noise_std, white_back)
        want.update(z_fine=zf, raw_fine=raw_f, rgb_fine=rgb_f, depth_fine=depth_f, opacity_fine=opac_f)
    # ---- one launch ----
    got = ops.render_fwd(rays, S, N, pc, pf if N > 0 else None, dtype, use_disp, perturb, pr, d["noise_coarse"], d["noise_fine"], noise_std,
                         white_back, u)
    torch.cuda.synchronize()
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), (k, (got[k] - want[k]).abs().max().item())
    # test_time form: no coarse colour / depth requested
    got2 = ops.render_fwd(rays, S, N, pc, pf if N > 0 else None, dtype, use_disp, perturb, pr, d["noise_coarse"], d["noise_fine"], noise_std,
                          white_back, u, want_coarse=False)
    assert "rgb_coarse" not in got2 and torch.equal(got2["opacity_coarse"], opac_c)
    if N > 0:
        assert torch.equal(got2["rgb_fine"], want["rgb_fine"])


@pytest.mark.parametrize("dtype,B,S,N,perturb,use_disp,noise_std,white_back,rand_u,kind", [c for c in CASES if c[3] > 0] + [
    ("bf16", 32768, 64, 128, 0.0, False, 0.0, True, False, "blender"),       # eval.py's chunk (eval.py:65), its flags (:69-79)
    ("fp32", 2048, 64, 128, 0.0, False, 0.0, True, False, "blender")])
def test_render_test_fwd_is_the_five_launches_of_test_time(dev, dtype, B, S, N, perturb, use_disp, noise_std, white_back, rand_u, kind):
    """test_time (rendering.py:209-213: the coarse model answers sigma_only, nerf.py:112-114) in ONE launch — nerfhip_render_test_fwd,
    whose coarse sub-passes run the network's sigma-only body — against the five launches render_rays issued for it until round 5,
    bit for bit: coarse depths, coarse sigma (B,S), coarse opacity, fine depths, fine rgb sigma, fine outputs."""
    from nerf_pl_amd import ops
    ms, _ = _models(dev, dtype)
    rays = O.make_rays(4, B, kind).to(dev)
    d = _draws(B, S, N, dev, seed=1)
    pr = d["perturb_rand"] if perturb > 0 else None
    u = d["u"] if rand_u else None
    pc, pf = ms[0].packed_weights(dtype), ms[1].packed_weights(dtype)
    z, sig_c = ops.mlp_fwd_rays_coarse(rays, S, pc, True, dtype, use_disp, perturb, pr)
    w_c, opac_c = ops.composite(sig_c, z, rays, d["noise_coarse"], noise_std, white_back)
    zf = ops.fine_z(z, w_c, N, u=u)
    raw_f = ops.mlp_fwd_rays(rays, zf, pf, False, dtype)
    _, opac_f, rgb_f, depth_f = ops.composite(raw_f, zf, rays, d["noise_fine"], noise_std, white_back)
    want = {"z_coarse": z, "raw_coarse": sig_c, "opacity_coarse": opac_c, "z_fine": zf, "raw_fine": raw_f, "rgb_fine": rgb_f,
            "depth_fine": depth_f, "opacity_fine": opac_f}
    got = ops.render_fwd(rays, S, N, pc, pf, dtype, use_disp, perturb, pr, d["noise_coarse"], d["noise_fine"], noise_std, white_back, u,
                         want_coarse=False, test_time=True)
    torch.cuda.synchronize()
    assert set(got) == set(want) and tuple(got["raw_coarse"].shape) == (B, S)
    for k in want:
        assert torch.equal(got[k], want[k]), (k, (got[k] - want[k]).abs().max().item())


def test_graph_renderer_single_launch_equals_its_launches(dev):
    """inference.GraphRenderer (eval.py's chunk loop as hipGraph replays): the captured chunk is ONE launch since round 6; same pixels
    as the captured five launches, ragged tail included."""
    from nerf_pl_amd.inference import GraphRenderer
    from nerf_pl_amd.models import rendering
    ms, emb = _models(dev, "bf16")
    rays = O.make_rays(21, 32768 + 1000, "blender").to(dev)
    out = {}
    prev = rendering.FUSE_TEST_TIME
    try:
        for on in (False, True):
            rendering.FUSE_TEST_TIME = on
            gr = GraphRenderer(ms, emb, 64, 128, False, True)
            out[on] = gr.render_to_host(rays, keys=("rgb_fine", "depth_fine"))
            torch.cuda.synchronize()
    finally:
        rendering.FUSE_TEST_TIME = prev
    for k in out[False]:
        assert torch.equal(out[False][k], out[True][k]), k


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("test_time", [False, True])
def test_render_rays_takes_the_single_launch_and_returns_the_same(dev, dtype, test_time):
    """models.rendering.render_rays under no_grad: same seed, single-launch kernel on / off -> the same dict, bit for bit, and the
    generator ends at the same offset (the four draws are made in the reference's order either way)."""
    from nerf_pl_amd import ops
    from nerf_pl_amd.models import rendering
    ms, emb = _models(dev, dtype)
    rays = O.make_rays(9, 512, "blender").to(dev)
    res, offs = {}, {}
    prev_tt = rendering.FUSE_TEST_TIME
    rendering.FUSE_TEST_TIME = True
    try:
        for on in (False, True):
            prev = ops.set_render_fused(on)
            try:
                torch.manual_seed(17)
                with torch.no_grad():
                    res[on] = rendering.render_rays(ms, emb, rays, 64, False, 1.0, 1.0, 128, 32768, True, test_time=test_time)
                offs[on] = torch.cuda.default_generators[dev.index or 0].get_offset()
            finally:
                ops.set_render_fused(prev)
    finally:
        rendering.FUSE_TEST_TIME = prev_tt
    assert offs[False] == offs[True]
    assert sorted(res[False]) == sorted(res[True])
    for k in res[False]:
        assert torch.equal(res[False][k], res[True][k]), k


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16_f8"])
@pytest.mark.parametrize("B,S,N,noise_std,white_back,perturb", [(1024, 64, 128, 0.0, True, 1.0), (32, 64, 64, 1.0, False, 0.0), (16, 64, 0, 1.0, True, 1.0)])
def test_training_forward_in_one_launch_is_the_four_launches(dev, dtype, B, S, N, noise_std, white_back, perturb, monkeypatch):
    """models/train_step.render_rays_train with the forward as ONE launch (nerfhip_render_train_fwd) and as the four launches of
    round 4: loss, PSNR, every rendered output and EVERY parameter gradient bit for bit (same saved activations, same d loss / d
    raw -> the backward cannot tell the difference).  (With the encodings saved in both: the bf16 one-launch step otherwise leaves them
    to the weight-gradient launch, whose split plan then weighs the jobs differently — another fp32 summation order; that form has
    its own tests, test_gpu_fused_step.py::test_bf16_step_without_saved_encodings_equals_the_step_with_them.)"""
    from nerf_pl_amd import ops
    from nerf_pl_amd.models import train_step
    from nerf_pl_amd.models.train_step import render_rays_train
    monkeypatch.setattr(train_step, "_regen_enc", False)
    rays = O.make_rays(4, B, "blender").to(dev)
    tgt = torch.rand(B, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    d = _draws(B, S, N, dev, seed=5)
    if N == 0:
        d.pop("u"), d.pop("noise_fine")
    outs = {}
    for on in (False, True):
        ms, emb = _models(dev, dtype)
        prev = ops.set_render_fused(on)
        try:
            assert ops.render_supported(B, S, N, dtype) == on
            res, loss, out3 = render_rays_train(ms, emb, rays, tgt, S, False, perturb, noise_std, N, white_back, draws=dict(d))
            loss.backward()
        finally:
            ops.set_render_fused(prev)
        torch.cuda.synchronize()
        outs[on] = (res, loss.detach().clone(), out3.clone(), [p.grad.clone() for m in ms[:2 if N > 0 else 1] for p in m.parameters()])
    (r0, l0, o0, g0), (r1, l1, o1, g1) = outs[False], outs[True]
    assert torch.equal(l0, l1) and torch.equal(o0, o1)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
    assert len(g0) == len(g1) == (48 if N > 0 else 24)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    # the arrival ticket is back at zero: a second launch reduces again
    assert int(ops._ticket(dev).sum().item()) == 0


def test_render_shapes_and_refusals(dev):
    from nerf_pl_amd import _lib, ops
    lib = _lib.load()
    ok = lambda B, S, N, dt: bool(lib.nerfhip_render_supported(B, S, N, ops.mlp_dtype_code(dt)))
    assert ok(1024, 64, 128, "bf16") and ok(32768, 64, 128, "bf16") and ok(17408, 64, 128, "bf16") and ok(4, 64, 0, "bf16")
    assert ok(4, 32, 32, "fp32") and not ok(4, 32, 32, "bf16")         # 4 x 32 = 128 points: one fp32 sub-pass, half a bf16 one
    assert not ok(1027, 64, 128, "bf16") and not ok(1024, 70, 128, "bf16") and not ok(1024, 64, 100, "bf16") and not ok(0, 64, 128, "bf16")
    # a shape the kernel refuses through the C ABI, and misaligned intermediates
    ms, _ = _models(dev, "bf16")
    a = _lib.RenderArgs()
    a.B, a.S_c, a.N_i = 1027, 64, 128
    assert lib.nerfhip_render_fwd(ctypes.addressof(a), ops.mlp_dtype_code("bf16"), None) != 0
    rays = O.make_rays(3, 8, "blender").to(dev)
    pc = ms[0].packed_weights("bf16")
    args, bufs, keep = ops._render_args(rays, 64, 0, pc, None, False, 0.0, None, None, None, 0.0, True, None, 1e-5, True)
    args.z_coarse = bufs["z_coarse"].data_ptr() + 4
    rc = lib.nerfhip_render_fwd(ctypes.addressof(args), ops.mlp_dtype_code("bf16"), _lib.stream_ptr())
    assert rc != 0 and b"align" in lib.nerfhip_error_string(rc).lower()
    # render_rays on a shape outside the kernel's: the launches, unchanged
    from nerf_pl_amd.models import rendering
    with torch.no_grad():
        out = rendering.render_rays(ms, _[0:2], O.make_rays(3, 10, "blender").to(dev), 64, False, 0, 0, 64, 32768, True)
    assert out["rgb_fine"].shape == (10, 3)
