"""accuracy of the in-kernel (bf16 path) encoding vs the oracle posenc, read back from the saved activation slabs"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import nerf_oracle as O
from nerf_pl_amd import ops
from nerf_pl_amd.models import NeRF
dev = torch.device('cuda:0')
m = NeRF(); m.load_state_dict(O.make_params(1)); m.mlp_dtype = 'bf16'; m = m.to(dev)
B, S = 256, 64
rays = O.make_rays(3, B, 'blender').to(dev)
z = torch.sort(2 + 4 * torch.rand(B, S, device=dev), -1)[0]
out_b = ops.mlp_fwd_rays(rays, z, m.packed_weights('bf16'), False, 'bf16')
m.mlp_dtype = 'fp32'
out_f = ops.mlp_fwd_rays(rays, z, m.packed_weights('fp32'), False, 'fp32')
print('bf16 vs fp32 kernel output: max abs diff rgb %.4g sigma %.4g (sigma scale %.3g)' % ((out_b[..., :3] - out_f[..., :3]).abs().max().item(),
      (out_b[..., 3] - out_f[..., 3]).abs().max().item(), out_f[..., 3].abs().max().item()))
