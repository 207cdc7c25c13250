import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from argparse import Namespace
from oracle import nerf_oracle as O
from nerf_pl_amd.system import GraphedTrainStep, NeRFSystem
dev = torch.device('cuda:0')
def run(perturb, graphed, load):
    hp = Namespace(N_samples=64, N_importance=64, use_disp=False, perturb=perturb, noise_std=0.0, chunk=1024 * 32, loss_type="mse",
                   lr=5e-4, weight_decay=0, decay_step=[100], decay_gamma=0.5, white_back=True)
    torch.manual_seed(0)
    system = NeRFSystem(hp)
    if load:
        system.nerf_coarse.load_state_dict(O.make_params(5, 4.0, 0.2)); system.nerf_fine.load_state_dict(O.make_params(6, 4.0, 0.2))
    for m in system.models: m.mlp_dtype = 'bf16'
    system = system.to(dev)
    (opt,), _ = system.configure_optimizers()
    st = GraphedTrainStep(system, opt, warmup=2 if graphed else 10**9)
    batch = {"rays": O.make_rays(1, 256, "blender").to(dev), "rgbs": torch.rand(256, 3, device=dev)}
    prev = system.nerf_fine.sigma.weight.detach().clone()
    out = []
    for i in range(6):
        o = st(batch)
        w = system.nerf_fine.sigma.weight.detach()
        g = system.nerf_fine.sigma.weight.grad
        out.append("%.2e/%.2e/%.4f" % ((w - prev).abs().max().item(), (g.abs().max().item() if g is not None else -1), o['loss'].item()))
        prev = w.clone()
    print("perturb", perturb, "graphed", graphed, "load", load, " dW/grad/loss per step:", out, flush=True)
for perturb in (0.0, 1.0):
    for graphed in (False, True):
        for load in (True, False):
            run(perturb, graphed, load)
