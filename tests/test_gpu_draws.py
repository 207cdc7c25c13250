"""GPU: the fused launches of round 4 against the launches they replace — every one bit for bit.

* nerfhip_torch_draws: the tensors torch.rand / torch.randn / torch.randint return for the same generator state, and the same
  generator state afterwards (the reference draws with those calls: rendering.py:203, :152, :39, :152; train.py:89-94);
* nerfhip_mlp_fwd_rays_coarse == nerfhip_sample_coarse_z -> nerfhip_mlp_fwd_rays (rendering.py:183-207);
* nerfhip_composite_train_fine_z == nerfhip_composite_train -> nerfhip_fine_z (rendering.py:143-172, :223-229);
* nerfhip_composite_train_loss == nerfhip_composite_train -> nerfhip_mse_psnr (losses.py:9-14, metrics.py:4-13).
The replaced launches are themselves pinned to the reference's golden vectors (tests/test_gpu_parity.py, test_rays.py)."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def _gen(dev):
    return torch.cuda.default_generators[dev.index]


@pytest.mark.parametrize("shape", [(1024, 64), (1024, 128), (1024, 192), (37, 70), (1, 1), (3000, 701), (5, 0)])
def test_rand_randn_equal_torch(dev, shape):
    from nerf_pl_amd import draws as D
    for kind, fn in (("rand", torch.rand), ("randn", torch.randn)):
        torch.manual_seed(1234)
        torch.rand(7, device=dev)                        # a generator that is not at offset 0
        off0 = _gen(dev).get_offset()
        want = fn(*shape, device=dev)
        off1 = _gen(dev).get_offset()
        torch.manual_seed(1234)
        torch.rand(7, device=dev)
        (got,) = D.draws([(kind, shape)], dev)
        assert _gen(dev).get_offset() == off1, (kind, shape, off0, off1, _gen(dev).get_offset())
        assert got.shape == want.shape
        assert torch.equal(got, want), (kind, shape, (got != want).sum().item(), (got - want).abs().max().item())


@pytest.mark.parametrize("n,high", [(1024, 20 * 200 * 200), (4096, 7), (100000, 2 ** 31 + 11), (1, 1), (5000, 2 ** 28 - 1), (5000, 2 ** 28),
                                    (700000, 2 ** 40 + 3)])
def test_randint_equals_torch(dev, n, high):
    from nerf_pl_amd import draws as D
    torch.manual_seed(99)
    want = torch.randint(0, high, (n,), device=dev)
    off1 = _gen(dev).get_offset()
    torch.manual_seed(99)
    (got,) = D.draws([("randint", (n,), high)], dev)
    assert _gen(dev).get_offset() == off1
    assert got.dtype == torch.int64 and torch.equal(got, want)


@pytest.mark.parametrize("perturb,noise_std,N", [(1.0, 0.0, 128), (1.0, 1.0, 64), (0.0, 1.0, 64), (0.0, 0.0, 0), (1.0, 0.0, 0)])
def test_step_draws_are_the_references_four_calls(dev, perturb, noise_std, N):
    """draws.step_draws == the torch calls of one render_rays, also when the noise tensors are not materialised."""
    from nerf_pl_amd import draws as D
    B, S = 257, 64
    torch.manual_seed(5)
    want = {}
    if perturb > 0:
        want["perturb_rand"] = torch.rand(B, S, device=dev)
    want["noise_coarse"] = torch.randn(B, S, device=dev)
    if N > 0:
        if perturb != 0:
            want["u"] = torch.rand(B, N, device=dev)
        want["noise_fine"] = torch.randn(B, S + N, device=dev)
    off = _gen(dev).get_offset()
    follow = torch.rand(3, device=dev)
    torch.manual_seed(5)
    got = D.step_draws(B, S, N, perturb, noise_std, dev)
    assert _gen(dev).get_offset() == off
    assert torch.equal(torch.rand(3, device=dev), follow)          # the stream goes on where torch's would
    for k, v in want.items():
        if k.startswith("noise") and noise_std == 0:
            assert k not in got
        else:
            assert torch.equal(got[k], v), k


def _store(dev, use_ndc=False):
    from nerf_pl_amd.rays import RayStore
    g = torch.Generator().manual_seed(8)
    n_img, hw = 5, 40
    c = torch.nn.functional.normalize(torch.randn(n_img, 3, generator=g), dim=-1) * 4.0
    poses = torch.cat([torch.linalg.qr(torch.randn(n_img, 3, 3, generator=g))[0], c[..., None]], -1).float().contiguous()
    rgbs = torch.rand(n_img * hw * hw, 3, generator=g)
    return RayStore(poses.to(dev), rgbs.to(dev), hw, hw, 55.0, 2.0, 6.0, use_ndc=use_ndc)


@pytest.mark.parametrize("use_ndc", [False, True])
def test_raystore_sample_is_randint_plus_sample_batch(dev, use_ndc):
    """RayStore.sample (one launch) == torch.randint -> nerfhip_sample_batch (round 3's two launches, the second one pinned to the
    reference's ray_utils.py by tests/test_rays.py), and with step_draws the batch carries the step's draws from the same stream."""
    from nerf_pl_amd import _lib
    from nerf_pl_amd._lib import check, ptr, stream_ptr
    st = _store(dev, use_ndc)
    B, S, N = 300, 64, 64
    torch.manual_seed(21)
    ids = torch.randint(0, st.n_pixels, (B,), device=dev)
    rays = torch.empty(B, 8, device=dev)
    rgbs = torch.empty(B, 3, device=dev)
    check(_lib.load().nerfhip_sample_batch(ptr(st.poses), ptr(ids), ptr(st.rgbs), B, st.H, st.W, st.focal, st.near, st.far,
                                           int(st.use_ndc), st.ndc_near_plane, ptr(rays), ptr(rgbs), stream_ptr()), "sample_batch")
    pr = torch.rand(B, S, device=dev)
    nc = torch.randn(B, S, device=dev)
    u = torch.rand(B, N, device=dev)
    nf = torch.randn(B, S + N, device=dev)
    off = _gen(dev).get_offset()
    torch.manual_seed(21)
    b = st.sample(B, step_draws=(S, N, 1.0, 1.0), return_ids=True)
    assert _gen(dev).get_offset() == off
    assert torch.equal(b["ids"], ids) and torch.equal(b["rays"], rays) and torch.equal(b["rgbs"], rgbs)
    for k, v in (("perturb_rand", pr), ("noise_coarse", nc), ("u", u), ("noise_fine", nf)):
        assert torch.equal(b["draws"][k], v), k
    torch.manual_seed(21)
    b2 = st.sample(B)
    assert torch.equal(b2["rays"], rays) and torch.equal(b2["rgbs"], rgbs) and "draws" not in b2


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_step_prologue_is_the_batch_launch_plus_the_pack_launch(dev, dtype):
    """nerfhip_train_prologue (RayStore.sample(..., pack_models=...)) == nerfhip_torch_draws + nerfhip_mlp_pack_weights_train_multi."""
    from nerf_pl_amd import ops
    st = _store(dev)
    models = _models(dev, dtype)
    B, S, N = 200, 64, 128
    want_packs = [(a.clone(), b.clone()) for a, b in ops.pack_models_train(models, dtype)]
    for m in models:
        for buf in m.train_buffers(dtype, dev):
            buf.zero_()
    torch.manual_seed(3)
    want = st.sample(B, step_draws=(S, N, 1.0, 1.0))
    off = _gen(dev).get_offset()
    torch.manual_seed(3)
    got = st.sample(B, step_draws=(S, N, 1.0, 1.0), pack_models=(models, dtype))
    assert _gen(dev).get_offset() == off and "packed" in got
    assert torch.equal(got["rays"], want["rays"]) and torch.equal(got["rgbs"], want["rgbs"])
    for k, v in want["draws"].items():
        assert torch.equal(got["draws"][k], v), k
    for m, (pf, pb) in zip(models, want_packs):
        a, b = m.train_buffers(dtype, dev)
        assert torch.equal(a, pf) and torch.equal(b, pb)


def test_captured_draws_walk_the_generator_stream(dev):
    """A hipGraph holding one draw launch: replay k returns what the k-th eager call would, and torch's generator ends where k
    eager calls would leave it (GraphDrawState: device-resident offset advanced by the kernel, generator moved by the host)."""
    from nerf_pl_amd import draws as D
    B, S, N = 128, 64, 128
    torch.manual_seed(77)
    want = []
    for _ in range(4):
        want.append((torch.rand(B, S, device=dev), torch.randn(B, S, device=dev), torch.rand(B, N, device=dev)))
        torch.randn(B, S + N, device=dev)
    off_end = _gen(dev).get_offset()
    torch.manual_seed(77)
    st = D.GraphDrawState(dev)
    st.arm()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with D.capturing(st), torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        outs = D.draws([("rand", (B, S)), ("randn", (B, S)), ("rand", (B, N)), ("randn", (B, S + N), False)], dev)
    assert outs[3] is None
    for k in range(4):
        st.before_replay()
        g.replay()
        st.after_replay()
        torch.cuda.synchronize()
        for a, b in zip(outs[:3], want[k]):
            assert torch.equal(a, b), k
    assert _gen(dev).get_offset() == off_end
    # an eager draw between replays moves the generator: the next replay follows it
    extra = torch.rand(5, device=dev)
    nxt = torch.rand(B, S, device=dev)
    torch.manual_seed(77)
    for _ in range(4):
        torch.rand(B, S, device=dev), torch.randn(B, S, device=dev), torch.rand(B, N, device=dev), torch.randn(B, S + N, device=dev)
    assert torch.equal(torch.rand(5, device=dev), extra)
    st.before_replay()
    g.replay()
    st.after_replay()
    torch.cuda.synchronize()
    assert torch.equal(outs[0], nxt)


def test_replica_self_check_and_torch_fallback(dev):
    """draws.replica_ok: the first use on a device checks the Philox replica against torch's own rand / randn / randint (values and
    generator offsets, two private generators) — it must hold on this torch build and leave the default generator untouched.
    With the replica declared broken (what the check does on a torch that moved its stream; NERFHIP_DRAWS=torch) every entry
    point makes the same tensors through torch's own calls: RayStore.sample with the step's draws and the weight images is bit
    for bit the replica's, the generator ends at the same offset."""
    from nerf_pl_amd import draws as D
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    torch.manual_seed(41)
    off0 = _gen(dev).get_offset()
    D._REPLICA.pop(idx, None)
    assert D._self_check(dev) is True
    assert D.replica_ok(dev) is True and D._REPLICA[idx] is True
    assert _gen(dev).get_offset() == off0            # private generators: the default stream has not moved
    st = _store(dev)
    models = _models(dev, "bf16")
    B, S, N = 300, 64, 128
    torch.manual_seed(9)
    want = st.sample(B, step_draws=(S, N, 1.0, 1.0), return_ids=True, pack_models=(models, "bf16"))
    want_packs = [tuple(b.clone() for b in m.train_buffers("bf16", dev)) for m in models]
    off = _gen(dev).get_offset()
    for m in models:
        for buf in m.train_buffers("bf16", dev):
            buf.zero_()
    try:
        D._REPLICA[idx] = False
        torch.manual_seed(9)
        got = st.sample(B, step_draws=(S, N, 1.0, 1.0), return_ids=True, pack_models=(models, "bf16"))
        assert _gen(dev).get_offset() == off
        assert torch.equal(got["ids"], want["ids"]) and torch.equal(got["rays"], want["rays"]) and torch.equal(got["rgbs"], want["rgbs"])
        assert set(got["draws"]) == set(want["draws"])
        for k, v in want["draws"].items():
            assert torch.equal(got["draws"][k], v), k
        for m, wp in zip(models, want_packs):
            for a, b in zip(m.train_buffers("bf16", dev), wp):
                assert torch.equal(a, b)
        # unread draws still advance the stream; the ids need not be returned
        torch.manual_seed(9)
        got2 = st.sample(B, step_draws=(S, N, 1.0, 0.0))
        assert _gen(dev).get_offset() == off and "ids" not in got2 and "noise_coarse" not in got2["draws"]
        assert torch.equal(got2["rays"], want["rays"]) and torch.equal(got2["draws"]["u"], want["draws"]["u"])
        assert not D.in_graph_stream(dev)
    finally:
        D._REPLICA[idx] = True


def test_modular_step_and_batch_source_share_one_captured_stream(dev):
    """(round-4 advisor finding) A captured step that mixes draws.py launches — the batch source, RayStore.sample — with the
    MODULAR render_rays (rendering.py's four rand / randn calls): inside the capture those four come from the same device-resident
    generator state as the batch's randint, so replay k consumes exactly what the k-th eager step would — torch's own
    capture-time bookkeeping would have restarted the render draws at the offset the batch's randint starts from (correlated
    pixel ids and jitter) and after_replay would have overwritten its advance."""
    from argparse import Namespace
    from nerf_pl_amd.system import GraphedTrainStep, NeRFSystem
    hp = Namespace(N_samples=64, N_importance=64, use_disp=False, perturb=1.0, noise_std=1.0, chunk=32768, loss_type="mse", lr=5e-4,
                   weight_decay=0, decay_step=[10 ** 9], decay_gamma=0.5, white_back=True, optimizer="adam", lr_scheduler="steplr")
    st = _store(dev)
    B, steps = 256, 7

    def run(graphed):
        system = NeRFSystem(hp)
        system.nerf_coarse.load_state_dict(O.make_params(5, 4.0, 0.2))
        system.nerf_fine.load_state_dict(O.make_params(6, 4.0, 0.2))
        for m in system.models:
            m.mlp_dtype = "fp32"
        system.fused_train_step = False                       # the modular graph: render_rays + MSELoss, torch-style draws
        system = system.to(dev)
        (opt,), _ = system.configure_optimizers()
        torch.manual_seed(123)
        losses = []
        if graphed:
            stepper = GraphedTrainStep(system, opt, warmup=2, batch_source=lambda: st.sample(B))
            for _ in range(steps):
                losses.append(stepper()["loss"].clone())
        else:
            for i in range(steps):
                out = system.training_step(st.sample(B), i)
                opt.zero_grad(set_to_none=True)
                out["loss"].backward()
                opt.step()
                losses.append(out["loss"].detach().clone())
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), _gen(dev).get_offset(), [p.detach().clone() for p in system.parameters()]

    l_e, off_e, p_e = run(False)
    l_g, off_g, p_g = run(True)
    assert off_e == off_g, (off_e, off_g)
    assert torch.equal(l_e, l_g), (l_e, l_g)
    for a, b in zip(p_e, p_g):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------- the fused launches
def _models(dev, dtype, seeds=(5, 6)):
    from helpers import build_models
    return build_models([O.make_params(seeds[0], 4.0, 0.2), O.make_params(seeds[1], 4.0, 0.2)], dev, dtype)[0]


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16_f8"])
@pytest.mark.parametrize("perturb,use_disp", [(1.0, False), (0.0, False), (0.5, True)])
def test_forward_with_its_own_coarse_depths(dev, dtype, perturb, use_disp):
    from nerf_pl_amd import ops
    B, S = 70, 64
    m = _models(dev, dtype)[0]
    rays = O.make_rays(3, B, "blender").to(dev)
    pr = torch.rand(B, S, device=dev) if perturb > 0 else None
    pk = m.packed_weights(dtype)
    z0 = ops.sample_coarse_z(rays, S, use_disp, perturb, pr)
    for save in (False, True):
        a0 = ops.alloc_acts(B * S, dtype, dev) if save else None
        a1 = ops.alloc_acts(B * S, dtype, dev) if save else None
        if save:
            a0.zero_(), a1.zero_()
        raw0 = ops.mlp_fwd_rays(rays, z0, pk, False, dtype, save=a0)
        z1, raw1 = ops.mlp_fwd_rays_coarse(rays, S, pk, False, dtype, use_disp, perturb, pr, save=a1)
        assert torch.equal(z0, z1) and torch.equal(raw0, raw1)
        if save:
            assert torch.equal(a0, a1)
    if dtype != "fp32":
        s0 = ops.mlp_fwd_rays(rays, z0, pk, True, dtype)
        z1, s1 = ops.mlp_fwd_rays_coarse(rays, S, pk, True, dtype, use_disp, perturb, pr)
        assert torch.equal(z0, z1) and torch.equal(s0, s1)


def _pass_inputs(dev, B, S, seed):
    g = torch.Generator().manual_seed(seed)
    raw = torch.randn(B, S, 4, generator=g)
    raw[..., :3] = torch.sigmoid(raw[..., :3])
    raw[..., 3] = raw[..., 3] * 3 + 1
    z = torch.sort(2 + 4 * torch.rand(B, S, generator=g), -1)[0]
    return (raw.to(dev), z.to(dev), O.make_rays(1, B, "blender").to(dev), torch.randn(B, S, generator=g).to(dev),
            torch.rand(B, 3, generator=g).to(dev))


@pytest.mark.parametrize("S,N", [(64, 128), (64, 64), (70, 33), (33, 128)])
@pytest.mark.parametrize("white_back,noise_std,rand_u", [(True, 0.0, True), (False, 1.0, False)])
def test_coarse_pass_compositing_with_fine_depths(dev, S, N, white_back, noise_std, rand_u):
    from nerf_pl_amd import ops
    B = 41
    raw, z, rays, noise, tgt = _pass_inputs(dev, B, S, S + N)
    u = torch.rand(B, N, device=dev) if rand_u else None
    gs = float(np.float32(2.0) / np.float32(3 * B))
    w, opac, rgb, depth, g_raw = ops.composite_train(raw, z, rays, noise, noise_std, white_back, tgt, gs)
    zf = ops.fine_z(z, w, N, u=u)
    for want_w in (False, True):
        w2, opac2, rgb2, depth2, g_raw2, zf2 = ops.composite_train_fine_z(raw, z, rays, noise, noise_std, white_back, tgt, gs, N, u=u,
                                                                          want_weights=want_w)
        assert (w2 is None) == (not want_w)
        for a, b in ((opac, opac2), (rgb, rgb2), (depth, depth2), (g_raw, g_raw2), (zf, zf2)) + (((w, w2),) if want_w else ()):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B", [1024, 37, 2, 1027])
@pytest.mark.parametrize("have_coarse", [True, False])
def test_last_pass_compositing_with_the_loss(dev, B, have_coarse):
    from nerf_pl_amd import ops
    S = 96
    raw, z, rays, noise, tgt = _pass_inputs(dev, B, S, B)
    rgb_c = torch.rand(B, 3, device=dev) if have_coarse else None
    gs = float(np.float32(2.0) / np.float32(3 * B))
    _, opac, rgb, depth, g_raw = ops.composite_train(raw, z, rays, None, 0.0, True, tgt, gs, want_weights=False)
    want = ops.mse_psnr_values(rgb_c if have_coarse else rgb, rgb if have_coarse else None, tgt)
    for _ in range(3):                                  # the ticket must be back at zero after every launch
        opac2, rgb2, depth2, g_raw2, out3 = ops.composite_train_loss(raw, z, rays, None, 0.0, True, tgt, gs, rgb_coarse=rgb_c)
        for a, b in ((opac, opac2), (rgb, rgb2), (depth, depth2), (g_raw, g_raw2), (want, out3)):
            assert torch.equal(a, b), (a, b)
