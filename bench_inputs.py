"""bench_inputs.py — constants and seeded synthetic inputs of bench.py (BASELINE.json: no dataset / checkpoint offline).

Split out of bench.py in round 6 (no behaviour change): the algorithmic FLOP / byte constants of SURVEY §8(a)/(d), the hardware peaks
of MI355X_MICROARCH.md, the seeded weights / rays / device-resident image stores, and `cpu_baseline()` — the ONLY function of the
bench that touches the CPU checker under oracle/ (test infrastructure; the product never imports it).
"""
import os
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))

FLOP_PER_POINT_FULL = 1186816      # 593,408 MAC, GEMMs only, no padding counted
FLOP_PER_POINT_DX = 1115392        # backward chain: 557,696 MAC (no dX into the encodings)
FLOP_PER_POINT_DW = 1186816        # weight-gradient GEMM: 593,408 MAC (the reference's autograd: what `frac` prices)
# ... of which the dW launch EXECUTES 527,872 MAC since round 6: xyz_encoding_final has no activation, so its two gradients follow
# from G = dY_dir^T h8 (the dir job, same size as before) by two small fp32 products per step instead of a 256 x 256 x P GEMM
# (csrc/mlp_layout.h kDwJobs).  Reported beside the algorithmic figure as `flops_executed` / `frac_mfma_executed_in_step`.
FLOP_PER_POINT_DW_EXECUTED = FLOP_PER_POINT_DW - 2 * 256 * 256
# The forward and the backward chain run the same fold (kernel layers: dir_encoding o xyz_encoding_final on h8 with the product matrix
# W_c = W_dir[:, :256] W_final, formed by the pack kernels): 256 x 256 MACs per point fewer each.  `frac` figures stay on the
# reference's FLOPs (what a step of the reference computes); the `*_executed` figures are what the MFMA pipe actually ran.
FLOP_PER_POINT_FULL_EXECUTED = FLOP_PER_POINT_FULL - 2 * 256 * 256
FLOP_PER_POINT_DX_EXECUTED = FLOP_PER_POINT_DX - 2 * 256 * 256
UNSAVED_SLABS_PER_TILE = 16        # slab slots of a 32-point tile block that nobody writes or reads: f in X, dL/df in dY (same source)
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
PEAK_TFLOPS = {"bf16": 2500.0, "bf16_f8": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md dense MFMA peaks of the forward / dX chain
PEAK_TFLOPS_FP8 = 5000.0           # ... and of the MX-scaled fp8 MFMA the dW GEMM of bf16_f8 runs on
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy)
FLOP_PER_RAY_EVAL = 64 * 982528 + 192 * 1186816      # test_time render: sigma-only coarse pass + full fine pass (290.7 M)
# `dtype` of the JSON line = the NARROWEST arithmetic inside the timed region
DTYPE_LABEL = {"bf16_f8": "bf16+fp8(dW)", "bf16": "bf16", "fp32": "f32"}


# Provenance of the parity claims beside the timing: what the -m gpu tests ASSERT (`bound`) and what the last
# full run of the suite measured (`measured`; profiles/r06_pytest_gpu_*.txt, profiles/r06_parity_errors.txt).  Static text: bench.py
# never runs a checker inside the timed run.
PARITY = {
    "source": "tests/ -m gpu at this round's HEAD; see DESIGN.md section 6",
    "bit_exact": ["searchsorted indices", "sample_coarse_z", "sample_pdf / fine_z cdf, indices and samples on the reference-minted (cdf, u) -> inds "
                  "triples under the ATen-order row total", "ray directions / NDC rays", "Philox draws vs torch.rand / randn / randint", "PFM bytes",
                  "every fused launch vs the launches it replaces"],
    "render_rays_fp32_vs_reference_golden": {"bound": "rtol 1e-4, absolute floor 1e-5", "measured_max_rel_above_floor": 1.6e-5,
                                             "test": "test_gpu_parity.py::test_render_rays_fp32_vs_reference_golden"},
    "render_rays_fp32_1024x192_vs_oracle": {"bound": "rtol 1e-4 / floor 1e-5 on every ray whose fine samples did not move; every outlier belongs to an identified moved ray; moved rays <= 5 %",
                                            "measured": "max rel 3.9e-6 on 988 unmoved rays; 36 moved rays, 0-3 of them outside 1e-4",
                                            "test": "test_gpu_parity.py::test_render_rays_fp32_benchmark_size_vs_oracle"},
    "gradients_fp32_vs_reference_48_tensors": {"coarse_model_and_fine_colour_branch": {"bound_of_max_abs_grad": 2e-4, "measured": 5.5e-5},
                                               "fine_trunk": {"bound_of_max_abs_grad": 2e-2, "measured": 8.8e-3,
                                                              "why": "clustered fine depths make the density gradient ill-conditioned in z on the reference's own "
                                                                     "arithmetic (tests/test_oracle_golden.py::test_fine_pass_conditioning)"},
                                               "fine_trunk_on_identical_depths": {"bound_of_max_abs_grad": 2e-3, "measured": 8.1e-4},
                                               "test": "test_gpu_training.py"},
    "gradients_bf16_vs_fp32_oracle": {"bound": "per-tensor cosine >= 0.99 (bf16), >= 0.98 (bf16_f8)", "test": "test_gpu_bf16.py"},
    "psnr_at_equal_steps_vs_reference_db": {"bound": "|mean paired difference| <= 0.10, standard error <= 0.05, steps 250-300, >= 8 live seeds of "
                                                     "tests/golden/reference_psnr_curves.json (the unmodified reference trained by oracle/make_psnr_curves.py)",
                                            "measured": {"fp32": "-0.016 +- 0.013 (14 seeds)", "bf16": "-0.067 +- 0.056 (14 seeds)"},
                                            "test": "test_gpu_psnr_gate.py::test_psnr_at_equal_steps_within_0p1_db_of_the_reference"},
    "psnr_at_equal_steps_vs_hip_fp32_db": {"bound": "|mean| <= 0.10, standard error <= 0.05 (16 live seeds, 31.3 dB plateau)",
                                           "measured": {"bf16": "+0.057 +- 0.042", "bf16_f8": "-0.007 +- 0.035"},
                                           "test": "test_gpu_psnr_gate.py::test_psnr_within_0p1_db_of_fp32_at_equal_steps"},
}


PARAM_SHAPES = [("xyz_encoding_%d.0" % (i + 1), 256, 63 if i == 0 else (319 if i == 4 else 256)) for i in range(8)] + \
               [("xyz_encoding_final", 256, 256), ("dir_encoding.0", 128, 283), ("sigma", 1, 256), ("rgb.0", 3, 128)]


def synth_params(seed, sigma_gain=1.0, sigma_bias=0.0):
    """nn.Linear's default U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from numpy PCG64 (identical on every rank and box);
    the density head is rescaled so that opacity is non-trivial (a trained-like field)."""
    import math

    import numpy as np
    rng = np.random.default_rng(seed)
    p = {}
    for name, fo, fi in PARAM_SHAPES:
        b = 1.0 / math.sqrt(fi)
        p[name + ".weight"] = torch.from_numpy(rng.uniform(-b, b, size=(fo, fi)).astype(np.float32))
        p[name + ".bias"] = torch.from_numpy(rng.uniform(-b, b, size=(fo,)).astype(np.float32))
    p["sigma.weight"] = p["sigma.weight"] * sigma_gain
    p["sigma.bias"] = p["sigma.bias"] * sigma_gain + sigma_bias
    return p


def synth_rays(seed, n):
    """Blender-style rays (n,8): origins (0,0,4)+0.1N, unit directions aimed near the scene centre, near 2, far 6
    (blender.py:34-35)."""
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(n, 3, generator=g)
    d = 0.8 * torch.randn(n, 3, generator=g) - o
    d = d / d.norm(dim=-1, keepdim=True)
    return torch.cat([o, d, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0)], 1).float().contiguous()


def cpu_baseline(B, S, N, seconds, train):
    """The pinned CPU oracle (torch-CPU restatement of the reference's render_rays, kind='port') timed
    on this node's host cores on the same workload shape; bounded to ~`seconds` of CPU work."""
    from oracle import nerf_oracle as O
    params = [O.make_params(0), O.make_params(1)]
    rays = O.make_rays(0, B, "blender")
    tgt = torch.rand(B, 3, generator=torch.Generator().manual_seed(0))
    rng = O.draw_rng(0, B, S, N, 1.0)
    if train:
        for d in params:
            for v in d.values():
                v.requires_grad_(True)
        opt = torch.optim.Adam([v for d in params for v in d.values()], lr=5e-4)

    def one(rays_, tgt_, rng_):
        if not train:
            with torch.no_grad():
                O.render_rays(params, rays_, S, False, 1.0, 0, N, True, False, rng=rng_)
            return
        res = O.render_rays(params, rays_, S, False, 1.0, 0, N, True, False, rng=rng_)
        loss = O.mse_loss(res, tgt_)
        opt.zero_grad()
        loss.backward()
        opt.step()

    ncpu = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))               # cores this process may run on (a container's cpuset)
    except AttributeError:
        usable = ncpu
    # torch-CPU oversubscribes badly on many-core hosts: pick the fastest of a few thread counts on a
    # 1/8-size probe, then time the full workload with it.
    best, best_t = 1, float("inf")
    Bp = max(32, B // 8)
    sub = {k: v[:Bp] for k, v in rng.items()}
    tried = sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128, ncpu)})
    probe_ms = {}
    for nt in tried:
        torch.set_num_threads(nt)
        one(rays[:Bp], tgt[:Bp], sub)
        t0 = time.perf_counter()
        one(rays[:Bp], tgt[:Bp], sub)
        t = time.perf_counter() - t0
        probe_ms[str(nt)] = round(t * 1e3, 1)
        if t < best_t:
            best, best_t = nt, t
    torch.set_num_threads(best)
    one(rays, tgt, rng)  # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        one(rays, tgt, rng)
        reps += 1
        if time.perf_counter() - t0 > seconds or reps >= 50:
            break
    dt = time.perf_counter() - t0
    what = "training step (fwd+loss+bwd+Adam)" if train else "render_rays fwd"
    # `cores` = the threads actually used (the contract's field); the node's own core count and what was tried ride beside it
    return {"value": round(B * reps / dt, 1), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "node_cores": ncpu, "usable_cores": usable, "threads": torch.get_num_threads(), "threads_probe_ms": probe_ms,
            "sample": "%d reps of the oracle's %s on %d rays x (%d+%d), torch-CPU fp32, %.1f s; %d threads = the fastest of %s on a 1/8-size "
                      "probe, node has %d cores (%d usable)" % (reps, what, B, S, N, dt, torch.get_num_threads(), tried, ncpu, usable)}


def synth_store(seed, dev, n_img=20, hw=200):
    """Device-resident synthetic training set in the reference's Blender layout (blender.py:42-69): camera poses on a
    radius-4 sphere looking at the origin + random pixel colours; batches are drawn and their rays generated on the GPU
    (nerf_pl_amd.rays.RayStore), so a training batch never crosses PCIe."""
    from nerf_pl_amd.rays import RayStore
    g = torch.Generator().manual_seed(seed)
    c = torch.nn.functional.normalize(torch.randn(n_img, 3, generator=g), dim=-1) * 4.0
    fwd = torch.nn.functional.normalize(-c, dim=-1)                       # camera looks down its -z axis at the origin
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.nn.functional.normalize(torch.cross(fwd, up, dim=-1), dim=-1)
    upv = torch.cross(right, fwd, dim=-1)
    poses = torch.stack([right, upv, -fwd, c], -1).float().contiguous()   # (n_img, 3, 4) = [R | t]
    rgbs = torch.rand(n_img * hw * hw, 3, generator=g)
    return RayStore(poses.to(dev), rgbs.to(dev), hw, hw, 0.5 * hw / 0.3, 2.0, 6.0)


def synth_store_ndc(seed, dev, n_img=20, hw=200):
    """The same in the reference's forward-facing LLFF layout (llff.py:236-253): cameras near the origin looking down -z with
    small rotations, rays converted to NDC (near plane 1.0), bounds 0..1, non-unit directions."""
    from nerf_pl_amd.rays import RayStore
    g = torch.Generator().manual_seed(seed)
    w = 0.1 * torch.randn(n_img, 3, generator=g)                          # small axis-angle rotations
    K = torch.zeros(n_img, 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -w[:, 2], w[:, 1], w[:, 2], -w[:, 0], -w[:, 1], w[:, 0]
    R = torch.matrix_exp(K)
    t = 0.3 * torch.randn(n_img, 3, 1, generator=g)
    poses = torch.cat([R, t], -1).float().contiguous()
    rgbs = torch.rand(n_img * hw * hw, 3, generator=g)
    return RayStore(poses.to(dev), rgbs.to(dev), hw, hw, 0.5 * hw / 0.35, 0.0, 1.0, use_ndc=True, ndc_near_plane=1.0)
