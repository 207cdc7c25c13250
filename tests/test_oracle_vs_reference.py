"""CPU, build container only: the oracle restatement against the REAL reference executed live (imported unmodified
through oracle/ref_shim.py) on fresh seeded inputs that are not in the golden file.  Skipped where /root/reference
does not exist (the GPU box)."""
import pytest
import torch

from oracle import nerf_oracle as O
from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


@pytest.mark.parametrize("seed,kind,N_i,white,test_time,disp", [(301, "blender", 64, True, False, False),
                                                               (302, "ndc", 32, False, True, False),
                                                               (303, "blender", 0, True, False, True)])
def test_render_rays_live_reference(seed, kind, N_i, white, test_time, disp):
    nerf, rend = ref_shim.load_reference()
    params = [O.make_params(seed, 6.0, 0.3), O.make_params(seed + 1, 6.0, 0.3)]
    models = []
    for p in params:
        m = nerf.NeRF()
        m.load_state_dict(p)
        models.append(m)
    rays = O.make_rays(seed, 24, kind)
    with torch.no_grad():                     # perturb=0, noise_std=0: the reference's randn draws are multiplied by 0
        ref = rend.render_rays(models, [nerf.Embedding(3, 10), nerf.Embedding(3, 4)], rays, 32, disp, 0, 0, N_i, 1024 * 32,
                               white, test_time=test_time)
        got = O.render_rays(params, rays, 32, disp, 0, 0, N_i, white, test_time)
    assert sorted(ref.keys()) == sorted(got.keys())
    for k in ref:
        assert torch.allclose(got[k], ref[k], rtol=1e-5, atol=1e-6), (k, (got[k] - ref[k]).abs().max().item())


@pytest.mark.parametrize("logscale", [True, False])
def test_embedding_live_reference(logscale):
    """Embedding.forward incl. the `logscale=False` bands torch.linspace(1, 2^(F-1), F) (nerf.py:16-19): bit-equal."""
    nerf, _ = ref_shim.load_reference()
    x = torch.randn(57, 3, generator=torch.Generator().manual_seed(5)) * 3
    for F in (10, 4, 6):
        assert torch.equal(O.posenc(x, F, logscale), nerf.Embedding(3, F, logscale=logscale)(x))


def test_ray_utils_live_reference():
    ru = ref_shim.load_reference_ray_utils()
    H, W, focal = 13, 17, 15.3
    c2w = O.make_pose(9)
    d = ru.get_ray_directions(H, W, focal)
    assert torch.equal(O.get_ray_directions(H, W, focal), d)
    ro, rd = ru.get_rays(d, c2w)
    oo, od = O.get_rays(d, c2w)
    assert torch.equal(oo, ro) and torch.allclose(od, rd, rtol=0, atol=1e-7)
    no, nd = ru.get_ndc_rays(H, W, focal, 1.0, ro, rd)
    po, pd = O.get_ndc_rays(H, W, focal, 1.0, ro, rd)
    assert torch.equal(po, no) and torch.equal(pd, nd)
