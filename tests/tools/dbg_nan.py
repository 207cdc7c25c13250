import torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import nerf_oracle as O
from tests.helpers import build_models
dev = torch.device('cuda:0')
for n in (1, 33, 300):
    g = torch.Generator().manual_seed(n)
    p = O.make_params(21, 3.0, 0.1)
    pts = torch.rand(n, 3, generator=g) * 4 - 2
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    x = torch.cat([O.posenc(pts, 10), O.posenc(dirs, 4)], 1)
    g_out = torch.randn(n, 4, generator=g)
    (m,), _ = build_models([p], dev, 'bf16')
    out = m(x.to(dev))
    (out * g_out.to(dev)).sum().backward()
    print('n', n, 'out nan', torch.isnan(out).sum().item())
    for name, prm in m.named_parameters():
        print('  ', name, 'nan', torch.isnan(prm.grad).sum().item(), 'of', prm.grad.numel(), 'absmax', prm.grad[~torch.isnan(prm.grad)].abs().max().item() if (~torch.isnan(prm.grad)).any() else None)
    b = m.xyz_encoding_1[0].bias.grad
    print('  nan bias rows', torch.isnan(b).nonzero().flatten().tolist())
    w = m.xyz_encoding_1[0].weight.grad
    print('  nan w cols of first nan row', torch.isnan(w[torch.isnan(b).nonzero().flatten()[0]]).nonzero().flatten().tolist() if torch.isnan(b).any() else None)
    big = (w.abs() > 5).nonzero()
    print('  big entries', big[:20].tolist(), w[w.abs()>5][:10].tolist())
