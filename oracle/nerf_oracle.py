"""CPU oracle for the NeRF volume-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This is a from-scratch CPU restatement (numpy + torch-CPU fp32) of the algorithm of
kwea123/nerf_pl's `models/nerf.py` + `models/rendering.py`.  It exists so that
the HIP kernels have something to be compared against on a machine where
`/root/reference` does not exist (the GPU box).  It is *pinned*: `oracle/make_golden.py`
runs the real reference (imported unmodified through `oracle/ref_shim.py`) and this
file on identical seeded inputs and commits the reference's outputs to `tests/golden/`;
`tests/test_oracle_golden.py` checks this file against those vectors on every run.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module.  The product (`nerf_pl_amd/`) never does and has no CPU fallback.

Every function cites the reference lines it follows (paths relative to /root/reference).
All RNG draws are *injected* as tensors so oracle and HIP path consume identical noise:
the reference's draw order inside one `render_rays` call is (rendering.py:203, :152,
:39, :152)  rand(B,S_c) [iff perturb>0] -> randn(B,S_c) -> rand(B,N_i) [iff perturb!=0]
-> randn(B,S_f).
"""
import math

import numpy as np
import torch

# ----------------------------------------------------------------------------- parameters
# state_dict layout of reference NeRF (models/nerf.py:60-81), default D=8 W=256 skips=[4]
XYZ_CH = 63
DIR_CH = 27
W = 256


def make_arch(D=8, W=W, N_freq_xyz=10, N_freq_dir=4, skips=(4,), logscale=True):
    """A NeRF / Embedding configuration (nerf.py:5-19, 42-57).  None / make_arch() = the reference's own (train.py:34-42)."""
    return dict(D=D, W=W, in_xyz=3 * (2 * N_freq_xyz + 1), in_dir=3 * (2 * N_freq_dir + 1), skips=tuple(skips),
                n_freq_xyz=N_freq_xyz, n_freq_dir=N_freq_dir, logscale=logscale)


def _arch(arch):
    return arch if arch is not None else make_arch()


def layer_shapes(D=8, Wd=W, in_xyz=XYZ_CH, in_dir=DIR_CH, skips=(4,)):
    """(name, out_features, in_features) in state_dict order. nerf.py:60-81."""
    shapes = []
    for i in range(D):
        if i == 0:
            fin = in_xyz
        elif i in skips:
            fin = Wd + in_xyz
        else:
            fin = Wd
        shapes.append((f"xyz_encoding_{i+1}.0", Wd, fin))
    shapes.append(("xyz_encoding_final", Wd, Wd))
    shapes.append(("dir_encoding.0", Wd // 2, Wd + in_dir))
    shapes.append(("sigma", 1, Wd))
    shapes.append(("rgb.0", 3, Wd // 2))
    return shapes


def make_params(seed, sigma_gain=1.0, sigma_bias=0.0, arch=None):
    """Deterministic, platform-independent parameters with nn.Linear's default
    distribution U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (numpy PCG64, not torch RNG, so the
    same weights can be rebuilt on any box from the seed alone).  `sigma_gain/bias`
    rescale the density head to emulate a trained (peaky) field."""
    rng = np.random.default_rng(seed)
    p = {}
    a = _arch(arch)
    for name, fo, fi in layer_shapes(a["D"], a["W"], a["in_xyz"], a["in_dir"], a["skips"]):
        b = 1.0 / math.sqrt(fi)
        p[name + ".weight"] = torch.from_numpy(rng.uniform(-b, b, size=(fo, fi)).astype(np.float32))
        p[name + ".bias"] = torch.from_numpy(rng.uniform(-b, b, size=(fo,)).astype(np.float32))
    p["sigma.weight"] = p["sigma.weight"] * sigma_gain
    p["sigma.bias"] = p["sigma.bias"] * sigma_gain + sigma_bias
    return p


def make_rays(seed, n, kind="blender"):
    """Seeded synthetic rays (B,8) = [o(3) d(3) near far].  SURVEY §8d.
    blender: o=(0,0,4)+0.1N, unit d aimed roughly at the origin, near 2 far 6 (blender.py:34-35).
    ndc    : forward-facing NDC-style rays, near 0 far 1, NON-unit d (ray_utils.py:75-92)."""
    g = torch.Generator().manual_seed(seed)
    if kind == "blender":
        o = torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(n, 3, generator=g)
        tgt = 0.8 * torch.randn(n, 3, generator=g)
        d = tgt - o
        d = d / d.norm(dim=-1, keepdim=True)
        near = torch.full((n, 1), 2.0)
        far = torch.full((n, 1), 6.0)
    elif kind == "ndc":
        o = torch.cat([torch.rand(n, 2, generator=g) * 2 - 1, -torch.ones(n, 1)], 1)
        d = torch.cat([0.3 * torch.randn(n, 2, generator=g), 2.0 + 0.2 * torch.rand(n, 1, generator=g)], 1)
        near = torch.zeros(n, 1)
        far = torch.ones(n, 1)
    else:
        raise ValueError(kind)
    return torch.cat([o, d, near, far], 1).float().contiguous()


# ----------------------------------------------------------------------------- a2: encoding
def posenc(x, n_freqs, logscale=True):
    """[x, sin(f_0 x), cos(f_0 x), sin(f_1 x), ...]  nerf.py:33-38; bands f_k = 2^k (logscale, nerf.py:17) or
    linspace(1, 2^(F-1), F) (nerf.py:19).
    Channel c: c<C -> x[c]; else k=(c-C)//(2C), sin if ((c-C)//C)%2==0 else cos."""
    x = x.float()
    n, C = x.shape
    out = torch.empty(n, C * (2 * n_freqs + 1), dtype=torch.float32)
    out[:, :C] = x
    col = C
    bands = 2 ** torch.linspace(0, n_freqs - 1, n_freqs) if logscale else torch.linspace(1, 2 ** (n_freqs - 1), n_freqs)
    for k in range(n_freqs):
        arg = x * bands[k]  # fp32 product first, then sin/cos
        out[:, col:col + C] = torch.sin(arg)
        out[:, col + C:col + 2 * C] = torch.cos(arg)
        col += 2 * C
    return out


# ----------------------------------------------------------------------------- a4: MLP
def _lin(p, name, h):
    return h @ p[name + ".weight"].t() + p[name + ".bias"]


def mlp_forward(p, x, sigma_only=False, return_acts=False, arch=None):
    """nerf.py:100-124.  x (n,90) [or (n,63) when sigma_only] -> (n,4)=[rgb,sigma] / (n,1)."""
    a = _arch(arch)
    c_xyz, c_dir = a["in_xyz"], a["in_dir"]
    enc_xyz = x[:, :c_xyz]
    h = enc_xyz
    acts = []
    for i in range(a["D"]):
        if i in a["skips"]:  # skip: [input_xyz, hidden]   nerf.py:108-109
            h = torch.cat([enc_xyz, h], -1)
        h = torch.relu(_lin(p, f"xyz_encoding_{i+1}.0", h))
        acts.append(h)
    sigma = _lin(p, "sigma", h)  # raw, no activation   nerf.py:112
    if sigma_only:
        return (sigma, acts) if return_acts else sigma
    feat = _lin(p, "xyz_encoding_final", h)  # no activation   nerf.py:116
    t = torch.relu(_lin(p, "dir_encoding.0", torch.cat([feat, x[:, c_xyz:c_xyz + c_dir]], -1)))
    rgb = torch.sigmoid(_lin(p, "rgb.0", t))
    out = torch.cat([rgb, sigma], -1)  # nerf.py:122
    return (out, acts) if return_acts else out


# ----------------------------------------------------------------------------- a9: searchsorted
def searchsorted_right(cdf, u):
    """Row-wise numpy.searchsorted(side='right'): first j with cdf[r,j] > u[r,k].
    Semantics of torchsearchsorted.searchsorted(cdf,u,side='right') (rendering.py:42).
    Pure numpy; int64 result (B,K)."""
    a = np.ascontiguousarray(cdf.detach().numpy() if torch.is_tensor(cdf) else cdf, dtype=np.float32)
    v = np.ascontiguousarray(u.detach().numpy() if torch.is_tensor(u) else u, dtype=np.float32)
    out = np.empty(v.shape, dtype=np.int64)
    for r in range(a.shape[0]):
        out[r] = np.searchsorted(a[r], v[r], side="right")
    return torch.from_numpy(out)


# ----------------------------------------------------------------------------- a8: sample_pdf
def _aten_multi_row_sum(load, nrows, size, zero):
    """ATen SumKernel.cpp multi_row_sum: `size` steps over `nrows` interleaved fp32 accumulators with a 4-level cascade
    (level step 16 below 2^20 steps).  load(i, k) -> the k-th row's term of step i."""
    f = np.float32
    num_levels = 4
    ceil_log2 = 0 if size <= 1 else int(size - 1).bit_length()
    level_power = max(4, ceil_log2 // num_levels)
    level_step = 1 << level_power
    level_mask = level_step - 1
    acc = [[zero.copy() for _ in range(nrows)] for _ in range(num_levels)]
    i = 0
    while i + level_step <= size:
        for _ in range(level_step):
            for k in range(nrows):
                acc[0][k] = (acc[0][k] + load(i, k)).astype(f)
            i += 1
        for j in range(1, num_levels):
            for k in range(nrows):
                acc[j][k] = (acc[j][k] + acc[j - 1][k]).astype(f)
                acc[j - 1][k] = zero.copy()
            if (i & (level_mask << (j * level_power))) != 0:
                break
    while i < size:
        for k in range(nrows):
            acc[0][k] = (acc[0][k] + load(i, k)).astype(f)
        i += 1
    for j in range(1, num_levels):
        for k in range(nrows):
            acc[0][k] = (acc[0][k] + acc[j][k]).astype(f)
    return acc[0]


def aten_row_total(row):
    """fp32 `torch.sum` of one contiguous row as ATen's CPU kernel orders the additions (torch 2.x SumKernel.cpp: cascade_sum ->
    vectorized_inner_sum -> row_sum).  The vectors are 8 floats wide on every x86 capability (checked here under AVX512, AVX2
    and DEFAULT dispatch: tests/test_oracle_golden.py), rows shorter than one vector take the same path with 1-float vectors:
    four interleaved vector accumulators over groups of four vectors, the left-over vectors into accumulator 0, accumulators
    1-3 added to 0, then a scalar chain  0 + row tail + the 8 vector lanes in order.  Restated because the reference's
    searchsorted indices (rendering.py:42) have knife edges on the LAST BIT of this total (rendering.py:30): the HIP kernels'
    NERFHIP_ROW_TOTAL_ATEN mode performs exactly these additions."""
    f = np.float32
    row = np.asarray(row, f)
    M = len(row)
    V = 8 if M >= 8 else 1
    vs = M // V
    n_ilp = vs // 4
    zero = np.zeros(V, f)

    def vec(i):
        return row[i * V:(i + 1) * V]
    ps = _aten_multi_row_sum(lambda i, k: vec(4 * i + k), 4, n_ilp, zero) if n_ilp > 0 else [zero.copy() for _ in range(4)]
    for i in range(n_ilp * 4, vs):
        ps[0] = (ps[0] + vec(i)).astype(f)
    for k in range(1, 4):
        ps[0] = (ps[0] + ps[k]).astype(f)
    if V == 1:
        return f(ps[0][0])
    acc = f(0)
    for k in range(vs * V, M):
        acc = f(acc + row[k])
    for k in range(V):
        acc = f(acc + ps[0][k])
    return acc


def pdf_to_cdf(weights, eps=1e-5, total_ulp=0, total="torch"):
    """rendering.py:29-33.  torch-CPU cumsum on fp32 (fp64 running sum rounded per element,
    SURVEY A.9) is kept because that IS the oracle's arithmetic.
    `total`: "torch" = torch.sum itself (the reference's line); "aten" = aten_row_total, the restatement of its addition
    order (bit-equal to "torch" on a CPU); "exact" = the correctly rounded sum (what the HIP kernels use by default).
    `total_ulp` shifts the fp32 row total by that many ulps: torch.sum's fp32 reduction order differs from the correctly
    rounded sum in the last bit, and the reference algorithm has knife edges (u == 1.0 in det
    mode, `denom < eps`) that flip on that last bit; tests of the default mode accept any shift in [-2, 2]."""
    w = weights.float() + eps
    if total == "torch":
        tot = torch.sum(w, -1, keepdim=True)
    elif total == "aten":
        tot = torch.from_numpy(np.array([aten_row_total(r) for r in w.contiguous().numpy()], np.float32))[:, None]
    elif total == "exact":
        tot = w.double().sum(-1, keepdim=True).float()
    else:
        raise ValueError(total)
    total = tot
    for _ in range(abs(int(total_ulp))):
        total = torch.nextafter(total, torch.full_like(total, math.inf if total_ulp > 0 else -math.inf))
    pdf = w / total
    cdf = torch.cumsum(pdf, -1)
    return torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)


def sample_pdf(bins, weights, n_importance, u=None, eps=1e-5, return_aux=False, total_ulp=0, total="torch"):
    """Inverse-CDF sampling, rendering.py:14-55.  u=None -> deterministic linspace (:36-37),
    else the injected (B,N_i) uniform draws (:39)."""
    B, M = weights.shape
    cdf = pdf_to_cdf(weights, eps, total_ulp, total)
    if u is None:
        u = torch.linspace(0, 1, n_importance).expand(B, n_importance)
    u = u.contiguous().float()
    inds = searchsorted_right(cdf, u)
    below = torch.clamp(inds - 1, min=0)  # :43
    above = torch.clamp(inds, max=M)  # :44
    cdf_b = torch.gather(cdf, 1, below)
    cdf_a = torch.gather(cdf, 1, above)
    bin_b = torch.gather(bins, 1, below)
    bin_a = torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)  # :51
    samples = bin_b + (u - cdf_b) / denom * (bin_a - bin_b)  # :54
    if return_aux:
        return samples, cdf, u, inds
    return samples


# ----------------------------------------------------------------------------- a5: z sampling
def coarse_z(rays, n_samples, use_disp=False, perturb=0.0, perturb_rand=None):
    """rendering.py:183-204."""
    near, far = rays[:, 6:7], rays[:, 7:8]
    t = torch.linspace(0, 1, n_samples)
    if not use_disp:
        z = near * (1 - t) + far * t
    else:
        z = 1 / (1 / near * (1 - t) + 1 / far * t)
    z = z.expand(rays.shape[0], n_samples)
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper = torch.cat([mid, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mid], -1)
        z = lower + (upper - lower) * (perturb * perturb_rand)
    return z.contiguous()


# ----------------------------------------------------------------------------- a7: compositing
def composite(sigmas, rgbs, z, rays_d, noise, white_back=False):
    """rendering.py:143-172.  sigmas (B,S) raw; rgbs (B,S,3) or None (weights only);
    noise (B,S) ALREADY multiplied by noise_std (or None == 0).
    Returns dict(weights, opacity[, rgb, depth])."""
    deltas = z[:, 1:] - z[:, :-1]
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1)  # :145-146
    deltas = deltas * torch.norm(rays_d.unsqueeze(1), dim=-1)  # :150
    s = sigmas if noise is None else sigmas + noise
    alphas = 1 - torch.exp(-deltas * torch.relu(s))  # :155
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)  # :156-157
    weights = alphas * torch.cumprod(shifted, -1)[:, :-1]  # :158-159
    out = {"weights": weights, "opacity": weights.sum(1)}
    if rgbs is not None:
        rgb = torch.sum(weights.unsqueeze(-1) * rgbs, -2)  # :166
        out["depth"] = torch.sum(weights * z, -1)  # :167
        if white_back:
            rgb = rgb + 1 - out["opacity"].unsqueeze(-1)  # :169-170
        out["rgb"] = rgb
    return out


# ----------------------------------------------------------------------------- a6/a10: render
def _infer(p, rays, z, dir_enc, noise, white_back, weights_only, arch=None):
    """`inference` closure, rendering.py:91-172 (point-chunk loop :125-133 collapsed: it is
    only a memory bound, results are chunk-invariant)."""
    B, S = z.shape
    xyz = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]  # :206-207 / :231-232
    a = _arch(arch)
    enc = posenc(xyz.reshape(-1, 3), a["n_freq_xyz"], a["logscale"])
    if weights_only:
        sig = mlp_forward(p, enc, sigma_only=True, arch=a).view(B, S)
        return composite(sig, None, z, rays[:, 3:6], noise, white_back)
    x = torch.cat([enc, dir_enc.repeat_interleave(S, 0)], 1)  # :119,:129
    o = mlp_forward(p, x, arch=a).view(B, S, 4)
    res = composite(o[..., 3], o[..., :3], z, rays[:, 3:6], noise, white_back)
    res["raw"] = o
    return res


def render_rays(params, rays, N_samples=64, use_disp=False, perturb=0, noise_std=1,
                N_importance=0, white_back=False, test_time=False, rng=None, return_aux=False, arch=None):
    """rendering.py:58-244 with injected RNG.  params = [coarse_dict, fine_dict].
    rng keys: 'perturb_rand' (B,S_c) U[0,1), 'noise_coarse' (B,S_c) N(0,1),
              'u' (B,N_i) U[0,1), 'noise_fine' (B,S_f) N(0,1)."""
    rng = rng or {}
    rays = rays.float()
    arch = _arch(arch)
    dir_enc = posenc(rays[:, 3:6], arch["n_freq_dir"], arch["logscale"])  # :186 (raw, possibly non-unit d)
    z = coarse_z(rays, N_samples, use_disp, perturb, rng.get("perturb_rand"))

    def nz(key):
        if noise_std == 0 or key not in rng:
            return None
        return rng[key] * noise_std  # :152

    aux = {"z_coarse": z}
    c = _infer(params[0], rays, z, dir_enc, nz("noise_coarse"), white_back, weights_only=test_time, arch=arch)
    if test_time:  # :209-213
        result = {"opacity_coarse": c["opacity"]}
    else:
        result = {"rgb_coarse": c["rgb"], "depth_coarse": c["depth"], "opacity_coarse": c["opacity"]}
    aux["weights_coarse"] = c["weights"]
    if N_importance > 0:  # :222-242
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        u = rng.get("u") if perturb != 0 else None  # det=(perturb==0)  :226
        z_new = sample_pdf(mid, c["weights"][:, 1:-1], N_importance, u=u).detach()  # :226
        zf, _ = torch.sort(torch.cat([z, z_new], -1), -1)  # :229
        f = _infer(params[1], rays, zf, dir_enc, nz("noise_fine"), white_back, weights_only=False, arch=arch)
        result["rgb_fine"] = f["rgb"]
        result["depth_fine"] = f["depth"]
        result["opacity_fine"] = f["opacity"]
        aux.update(z_new=z_new, z_fine=zf, weights_fine=f["weights"])
    return (result, aux) if return_aux else result


def draw_rng(seed, B, S_c, N_i, perturb):
    """Seeded RNG tensors in the reference's consumption order."""
    g = torch.Generator().manual_seed(seed)
    r = {}
    if perturb > 0:
        r["perturb_rand"] = torch.rand(B, S_c, generator=g)
    r["noise_coarse"] = torch.randn(B, S_c, generator=g)
    if N_i > 0:
        if perturb != 0:
            r["u"] = torch.rand(B, N_i, generator=g)
        r["noise_fine"] = torch.randn(B, S_c + N_i, generator=g)
    return r


# ----------------------------------------------------------------------------- a12: loss
def mse_loss(result, target):
    """losses.py:9-14."""
    loss = torch.mean((result["rgb_coarse"] - target) ** 2)
    if "rgb_fine" in result:
        loss = loss + torch.mean((result["rgb_fine"] - target) ** 2)
    return loss


def psnr(pred, gt):
    """metrics.py:4-13."""
    return -10 * torch.log10(torch.mean((pred - gt) ** 2))


def matches_some_total_rounding(got, bins, weights, n_importance, u=None, atol=2e-6, eps=1e-5):
    """Element-wise: does `got` equal sample_pdf(...) for SOME rounding (within +-2 ulp) of the fp32
    row total?  Returns a bool tensor.  See pdf_to_cdf."""
    ok = torch.zeros_like(got, dtype=torch.bool)
    for s in (-2, -1, 0, 1, 2):
        ok |= (got - sample_pdf(bins, weights, n_importance, u=u, eps=eps, total_ulp=s)).abs() <= atol
    return ok


def grad_digest(g):
    """Compact fingerprint of a gradient tensor for golden files: [sum, l2, first 8, last 8]."""
    f = g.reshape(-1).double()
    head = f[:8]
    tail = f[-8:]
    if head.numel() < 8:
        head = torch.cat([head, torch.zeros(8 - head.numel(), dtype=torch.float64)])
        tail = torch.cat([tail, torch.zeros(8 - tail.numel(), dtype=torch.float64)])
    return torch.cat([f.sum()[None], f.norm()[None], head, tail]).float()


# ----------------------------------------------------------------------------- ray geometry (N1)
# Restatement of reference datasets/ray_utils.py (pinned by tests/golden: "rg_*" vectors minted from the real file).
def get_ray_directions(H, W, focal):
    """datasets/ray_utils.py:5-24: pixel (i = column, j = row) -> camera-space direction
    ((i - W/2)/focal, -(j - H/2)/focal, -1), no +0.5 pixel centring.  (H, W, 3) fp32."""
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    return torch.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -torch.ones_like(i)], -1)


def get_rays(directions, c2w):
    """datasets/ray_utils.py:27-52: rotate by c2w[:, :3], normalise, origin = c2w[:, 3].  -> (H*W,3) o, (H*W,3) d."""
    rays_d = directions @ c2w[:, :3].T
    rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    rays_o = c2w[:, 3].expand(rays_d.shape)
    return rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)


def get_ndc_rays(H, W, focal, near, rays_o, rays_d):
    """datasets/ray_utils.py:55-94: shift origins to the near plane, project to NDC."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    ox_oz = rays_o[..., 0] / rays_o[..., 2]
    oy_oz = rays_o[..., 1] / rays_o[..., 2]
    o0 = -1. / (W / (2. * focal)) * ox_oz
    o1 = -1. / (H / (2. * focal)) * oy_oz
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - ox_oz)
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - oy_oz)
    d2 = 1 - o2
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def make_pose(seed):
    """A deterministic camera-to-world (3,4): random rotation (QR of a PCG64 gaussian matrix, det +1) and an origin
    ~4 units from the scene centre, like the Blender cameras."""
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    t = rng.standard_normal(3)
    t = 4.0 * t / np.linalg.norm(t)
    return torch.from_numpy(np.concatenate([q, t[:, None]], 1).astype(np.float32))
