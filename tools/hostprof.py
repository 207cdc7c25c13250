"""Host-side cost of one training step: cProfile over 200 steps (GPU work is async; this shows Python/launch overhead)."""
import cProfile, pstats, sys, os, io
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _synth  # tools/_synth.py
from nerf_pl_amd.system import NeRFSystem
dev = torch.device("cuda:0")
hp = Namespace(N_samples=64, N_importance=128, use_disp=False, perturb=1.0, noise_std=0.0, chunk=1024 * 32, loss_type="mse",
               lr=5e-4, weight_decay=0, decay_step=[2, 4, 8], decay_gamma=0.5, white_back=True)
system = NeRFSystem(hp)
for m in system.models:
    m.mlp_dtype = "bf16"
system = system.to(dev)
(opt,), _ = system.configure_optimizers()
batch = {"rays": _synth.make_rays(1, 1024, dev), "rgbs": torch.rand(1024, 3, device=dev)}
def step():
    out = system.training_step(batch, 0)
    opt.zero_grad(set_to_none=True)
    out["loss"].backward()
    opt.step()
for _ in range(20): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host issue %.3f ms/step, wall %.3f ms/step" % (t_issue / 200 * 1e3, t_all / 200 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
