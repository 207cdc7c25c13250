import torch  # noqa: F401

from oracle import nerf_oracle as O
from oracle.replay import ReplayRNG, fused_draws, hip_render  # noqa: F401  (checker plumbing lives under oracle/, not in the test tree)
from oracle.scenes import analytic_field, analytic_scene, brick_field, brick_scene  # noqa: F401


def case_from_golden(golden, name, prefix="rr"):
    key = f"{prefix}_{name}_cfg" if prefix == "rr" else f"{prefix}_cfg"
    cfg = golden[key].tolist()
    kind = {0: "blender", 1: "ndc"}[int(cfg[0])]
    B, S_c, N_i = int(cfg[1]), int(cfg[2]), int(cfg[3])
    kw = dict(N_samples=S_c, use_disp=bool(cfg[4]), perturb=cfg[5], noise_std=cfg[6], N_importance=N_i,
              white_back=bool(cfg[7]), test_time=bool(cfg[8]))
    sg, sb, seed = cfg[9], cfg[10], int(cfg[11])
    params = [O.make_params(seed, sg, sb), O.make_params(seed + 500, sg, sb)]
    rays = O.make_rays(seed, B, kind)
    rng = O.draw_rng(seed, B, S_c, N_i, cfg[5])
    return params, rays, kw, rng


def build_models(params, device, dtype="fp32"):
    from nerf_pl_amd.models import Embedding, NeRF
    ms = []
    for p in params:
        m = NeRF()
        m.load_state_dict(p)
        m.mlp_dtype = dtype
        ms.append(m.to(device))
    return ms, [Embedding(3, 10), Embedding(3, 4)]




def build_arch_models(arch, seeds, device, dtype="fp32", sigma_gain=6.0, sigma_bias=0.3):
    """NeRF models + embeddings of a non-default configuration (oracle.make_arch) with the oracle's seeded weights"""
    from nerf_pl_amd.models import Embedding, NeRF
    ms, params = [], []
    for sd in seeds:
        p = O.make_params(sd, sigma_gain, sigma_bias, arch=arch)
        m = NeRF(D=arch["D"], W=arch["W"], in_channels_xyz=arch["in_xyz"], in_channels_dir=arch["in_dir"], skips=list(arch["skips"]))
        m.load_state_dict(p)
        m.mlp_dtype = dtype
        ms.append(m.to(device))
        params.append(p)
    embs = [Embedding(3, arch["n_freq_xyz"], logscale=arch["logscale"]), Embedding(3, arch["n_freq_dir"], logscale=arch["logscale"])]
    return ms, embs, params
