"""Debug: decode saved activations / gates / dY slabs of the bf16 training path and cross-check them."""
import ctypes, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import nerf_oracle as O
from tests.helpers import build_models
from nerf_pl_amd import ops, _lib
from nerf_pl_amd._lib import ptr, stream_ptr
dev = torch.device('cuda:0')
lib = _lib.load()
KACT, KDY, NMASK = 158, 156, 9
n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
g = torch.Generator().manual_seed(n)
p = O.make_params(21, 3.0, 0.1)
pts = torch.rand(n, 3, generator=g) * 4 - 2
dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
x = torch.cat([O.posenc(pts, 10), O.posenc(dirs, 4)], 1).to(dev)
g_out = torch.randn(n, 4, generator=g).to(dev)
(m,), _ = build_models([p], dev, 'bf16')
acts = ops.alloc_acts(n, 'bf16', dev)
acts.fill_(0xEE)
out = ops.mlp_fwd_embedded(x, m.packed_weights('bf16'), False, 'bf16', save=acts)
torch.cuda.synchronize()
tile_bytes = KACT * 1024 + NMASK * 1024
tiles = acts.numel() // tile_bytes
A = acts.view(tiles, tile_bytes)
slabs = A[:, :KACT * 1024].reshape(tiles, KACT, 64, 8, 2).contiguous().view(torch.bfloat16).reshape(tiles, KACT, 64, 8).float()
gates = A[:, KACT * 1024:].reshape(tiles, NMASK, 64, 4, 4).contiguous().view(torch.int32).reshape(tiles, NMASK, 64, 4)
def gate_bits(piece, nslab):
    w = gates[:, piece]                      # (tiles, 64, 4) int32
    bits = torch.zeros(tiles, nslab, 64, 8, dtype=torch.bool, device=dev)
    for ks in range(nslab):
        for j in range(8):
            idx = 8 * ks + j
            bits[:, ks, :, j] = ((w[:, :, idx >> 5] >> (16 * (idx & 1) + 15 - ((idx & 31) >> 1))) & 1).bool()   # mlp_layout.h gate_bit
    return bits
for l in range(1, 9):
    h = slabs[:, 6 + 16 * (l - 1): 6 + 16 * l]          # (tiles,16,64,8)
    gb = gate_bits(l - 1, 16)
    mism = ((h > 0) != gb).sum().item()
    print('layer', l, 'gate vs h>0 mismatches', mism, 'of', gb.numel(), 'h nan', torch.isnan(h).sum().item())
t = slabs[:, 150:158]
print('t gate mism', ((t > 0) != gate_bits(8, 8)).sum().item())
# backward
code = 1
dys = torch.empty(int(lib.nerfhip_mlp_dy_bytes(n, code)), device=dev, dtype=torch.uint8); dys.fill_(0xEE)
ws = torch.empty(int(lib.nerfhip_mlp_dw_workspace_bytes(n, code)), device=dev, dtype=torch.uint8)
gw = [torch.zeros(s, device=dev) for s in ops.PARAM_SHAPES]; gb_ = [torch.zeros(s[0], device=dev) for s in ops.PARAM_SHAPES]
gwp = (ctypes.c_void_p * 12)(*[t_.data_ptr() for t_ in gw]); gbp = (ctypes.c_void_p * 12)(*[t_.data_ptr() for t_ in gb_])
rc = lib.nerfhip_mlp_bwd(ptr(g_out), ptr(out), n, ptr(m.packed_weights_bwd('bf16')), ptr(acts), ptr(dys), ptr(ws), gwp, gbp, 0, code, stream_ptr())
torch.cuda.synchronize()
print('rc', rc)
D = dys.view(tiles, KDY, 64, 8, 2).contiguous().view(torch.bfloat16).reshape(tiles, KDY, 64, 8).float()
for l in range(1, 9):
    sec = 28 + 16 * (8 - l)
    d = D[:, sec:sec + 16]
    print('dY layer', l, 'nan', torch.isnan(d).sum().item(), 'absmax', d[~torch.isnan(d)].abs().max().item(),
          'nonzero where gate closed', ((d != 0) & ~gate_bits(l - 1, 16)).sum().item())
    if l == 1:
        bad = (d.abs() > 1) | torch.isnan(d)
        print('   bad idx (tile,ks,lane,j):', bad.nonzero()[:12].tolist())
print('gW1 nan', torch.isnan(gw[0]).sum().item(), 'absmax', gw[0][~torch.isnan(gw[0])].abs().max().item())
