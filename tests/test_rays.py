"""N1 ray generation: the oracle restatement against the real reference's datasets/ray_utils.py outputs (CPU), and the
HIP kernels against the same golden vectors and the oracle (GPU)."""
import pytest
import torch

from oracle import nerf_oracle as O


def _cfg(golden, tag):
    H, W, focal, seed = golden[f"rg_{tag}_cfg"].tolist()
    return int(H), int(W), float(focal), int(seed)


@pytest.mark.parametrize("tag", ["blender", "llff"])
def test_oracle_ray_geometry_vs_reference_golden(golden, tag):
    H, W, focal, seed = _cfg(golden, tag)
    dirs = O.get_ray_directions(H, W, focal)
    assert torch.equal(dirs, golden[f"rg_{tag}_dirs"])
    ro, rd = O.get_rays(dirs, O.make_pose(seed))
    assert torch.equal(ro, golden[f"rg_{tag}_o"])
    assert torch.allclose(rd, golden[f"rg_{tag}_d"], rtol=0, atol=1e-7)
    assert torch.allclose(rd.norm(dim=-1), torch.ones(H * W), atol=1e-6)
    if tag == "llff":
        no, nd = O.get_ndc_rays(H, W, focal, 1.0, golden["rg_llff_o"], golden["rg_llff_d"])
        assert torch.equal(no, golden["rg_llff_ndc_o"]) and torch.equal(nd, golden["rg_llff_ndc_d"])
        # NDC property (ray_utils.py:60-61): origins sit on the near plane z = -1, o + d reaches the far plane z = +1
        assert torch.allclose(no[:, 2], -torch.ones(H * W), atol=1e-5)
        assert torch.allclose(no[:, 2] + nd[:, 2], torch.ones(H * W), atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["blender", "llff"])
def test_hip_ray_geometry_vs_reference_golden(golden, dev, tag):
    from nerf_pl_amd import rays as R
    H, W, focal, seed = _cfg(golden, tag)
    c2w = O.make_pose(seed)
    dirs = R.get_ray_directions(H, W, focal, device=dev)
    assert torch.equal(dirs.cpu(), golden[f"rg_{tag}_dirs"])                      # exact: same fp32 ops
    ro, rd = R.get_rays(dirs, c2w.to(dev))
    assert torch.equal(ro.cpu(), golden[f"rg_{tag}_o"])
    assert torch.allclose(rd.cpu(), golden[f"rg_{tag}_d"], rtol=0, atol=2e-7)     # 3-term dot: summation order / fma
    near, far = (2.0, 6.0) if tag == "blender" else (0.0, 1.0)
    fused = R.gen_rays(c2w.to(dev), H, W, focal, near, far, use_ndc=(tag == "llff"), ndc_near_plane=1.0)
    assert fused.shape == (H * W, 8)
    if tag == "llff":
        no, nd = R.get_ndc_rays(H, W, focal, 1.0, golden["rg_llff_o"].to(dev), golden["rg_llff_d"].to(dev))
        assert torch.equal(no.cpu(), golden["rg_llff_ndc_o"]) and torch.equal(nd.cpu(), golden["rg_llff_ndc_d"])
        assert torch.allclose(fused[:, :3].cpu(), golden["rg_llff_ndc_o"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(fused[:, 3:6].cpu(), golden["rg_llff_ndc_d"], rtol=1e-5, atol=1e-6)
    else:
        assert torch.equal(fused[:, :3].cpu(), golden["rg_blender_o"])
        assert torch.allclose(fused[:, 3:6].cpu(), golden["rg_blender_d"], rtol=0, atol=2e-7)
    assert torch.equal(fused[:, 6].cpu(), torch.full((H * W,), near)) and torch.equal(fused[:, 7].cpu(), torch.full((H * W,), far))


@pytest.mark.gpu
def test_ray_store_batches_match_full_image_rays(dev):
    """RayStore.sample draws pixel ids on the device; its rays/rgbs equal rows of the per-image tables (multi-image ids)."""
    from nerf_pl_amd import rays as R
    H, W, focal = 17, 23, 19.25
    poses = torch.stack([O.make_pose(s) for s in (1, 2, 3)]).to(dev)
    rgbs = torch.rand(3 * H * W, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    store = R.RayStore(poses, rgbs, H, W, focal, 2.0, 6.0)
    full = torch.cat([store.image_rays(i) for i in range(3)], 0)
    for i in range(3):
        d = O.get_ray_directions(H, W, focal)
        ro, rd = O.get_rays(d, poses[i].cpu())
        assert torch.equal(full[i * H * W:(i + 1) * H * W, :3].cpu(), ro)
        assert torch.allclose(full[i * H * W:(i + 1) * H * W, 3:6].cpu(), rd, rtol=0, atol=2e-7)
    g = torch.Generator(device=dev).manual_seed(5)
    batch = store.sample(4096, generator=g)
    ids = torch.randint(0, len(store), (4096,), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    assert torch.equal(batch["rays"], full[ids]) and torch.equal(batch["rgbs"], rgbs[ids])
    assert R.gen_rays(poses, H, W, focal, 2.0, 6.0, pixel_ids=torch.zeros(0, dtype=torch.int64, device=dev)).shape == (0, 8)


@pytest.mark.gpu
def test_train_from_ray_store_end_to_end(dev):
    """N1 + path + N2 together: batches drawn and their rays generated on the GPU (RayStore), rendered and trained by the
    HIP path with the whole step replayed as a hipGraph; the loss on a fixed probe set goes down."""
    from argparse import Namespace
    from nerf_pl_amd import rays as R
    from nerf_pl_amd.inference import batched_inference
    from nerf_pl_amd.models import Embedding, NeRF
    from nerf_pl_amd.system import GraphedTrainStep, NeRFSystem
    H = W = 32
    focal = 40.0
    poses = torch.stack([O.make_pose(s) for s in range(6)]).to(dev)
    teacher = []
    for seed in (777, 778):
        m = NeRF()
        p = O.make_params(seed, 30.0, 0.0)
        p["rgb.0.weight"] = p["rgb.0.weight"] * 30.0
        m.load_state_dict(p)
        m.mlp_dtype = "bf16"
        teacher.append(m.to(dev))
    emb = [Embedding(3, 10), Embedding(3, 4)]
    all_rays = R.gen_rays(poses, H, W, focal, 2.0, 6.0)
    rgbs = batched_inference(teacher, emb, all_rays, 64, 64, False, 32768, True)["rgb_fine"]
    store = R.RayStore(poses, rgbs, H, W, focal, 2.0, 6.0)
    hp = Namespace(N_samples=64, N_importance=64, use_disp=False, perturb=1.0, noise_std=0.0, chunk=1024 * 32, loss_type="mse",
                   lr=5e-4, weight_decay=0, decay_step=[10 ** 6], decay_gamma=0.5, white_back=True)
    system = NeRFSystem(hp)
    # seeded init with a positive density bias: an unseeded default init can start with relu(sigma) == 0 on every
    # sample (white image, exactly zero gradients) — the classic dead-density start, which never trains
    system.nerf_coarse.load_state_dict(O.make_params(5, 4.0, 0.2))
    system.nerf_fine.load_state_dict(O.make_params(6, 4.0, 0.2))
    for m in system.models:
        m.mlp_dtype = "bf16"
    system = system.to(dev)
    (opt,), _ = system.configure_optimizers()
    stepper = GraphedTrainStep(system, opt, warmup=2)
    gen = torch.Generator(device=dev).manual_seed(0)

    def probe():
        with torch.no_grad():
            pred = batched_inference(system.models, emb, all_rays[:2048], 64, 64, False, 32768, True)["rgb_fine"]
        return torch.mean((pred - rgbs[:2048]) ** 2).item()

    before = probe()
    for _ in range(60):
        stepper(store.sample(512, generator=gen))
    after = probe()
    assert stepper.graph is not None
    assert after < 0.8 * before, (before, after)
