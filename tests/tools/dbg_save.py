import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import nerf_oracle as O
from nerf_pl_amd import ops
from nerf_pl_amd.models import NeRF
dev = torch.device('cuda:0')
def mk(seed):
    m = NeRF(); m.load_state_dict(O.make_params(seed, 4.0, 0.2)); m.mlp_dtype = 'bf16'; return m.to(dev)
mc, mf = mk(100), mk(101)
B, S, N = 1024, 64, 128
rays = O.make_rays(1234, B, 'blender').to(dev)
def t(fn, reps=10):
    for _ in range(2): fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); e[0].record()
    for _ in range(reps): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) * 1e3 / reps
pk = mf.packed_weights()
zr = torch.sort(2 + 4 * torch.rand(B, S + N, device=dev), -1)[0]
z = ops.sample_coarse_z(rays, S, False, 1.0, torch.rand(B, S, device=dev))
raw = ops.mlp_fwd_rays(rays, z, mc.packed_weights(), False, 'bf16')
w, op, rgb, dep = ops.composite(raw, z, rays, None, 0.0, True)
zf = ops.fine_z(z, w, N, u=torch.rand(B, N, device=dev))
acts = ops.alloc_acts(B * (S + N), 'bf16', dev)
print('random z, fixed acts      : %.1f us' % t(lambda: ops.mlp_fwd_rays(rays, zr, pk, False, 'bf16', save=acts)))
print('fine_z z, fixed acts      : %.1f us' % t(lambda: ops.mlp_fwd_rays(rays, zf, pk, False, 'bf16', save=acts)))
print('fine_z z, no save         : %.1f us' % t(lambda: ops.mlp_fwd_rays(rays, zf, pk, False, 'bf16')))
def fresh():
    a = ops.alloc_acts(B * (S + N), 'bf16', dev)
    ops.mlp_fwd_rays(rays, zf, pk, False, 'bf16', save=a)
print('fine_z z, fresh acts      : %.1f us' % t(fresh))
print('zf stats: min dz %.3g, frac dz<1e-4 %.3f' % ((zf[:,1:]-zf[:,:-1]).min().item(), ((zf[:,1:]-zf[:,:-1])<1e-4).float().mean().item()))
# with autograd path (what training does)
from nerf_pl_amd.models.mlp_autograd import mlp_rays
def auto():
    with torch.enable_grad():
        o = mlp_rays(mf, rays, zf, False)
    return o
print('autograd fwd (fine_z)     : %.1f us' % t(auto))
