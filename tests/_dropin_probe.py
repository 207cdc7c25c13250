"""Runs in a SUBPROCESS (tests/test_dropin_reference_scripts.py): imports the REAL /root/reference/train.py and eval.py with
the hot-path modules replaced by `nerf_pl_amd.install()` and the out-of-scope dependencies that this image lacks
(pytorch_lightning 0.7.5, kornia, torchvision, cv2, imageio, the dataset loaders) stubbed, and prints a JSON report."""
import importlib.util
import inspect
import json
import os
import sys
import types

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)

import torch  # noqa: E402


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ---- out-of-scope dependencies (not on the hot path) ------------------------------------------------------
class _LightningModule(torch.nn.Module):
    pass


stub("pytorch_lightning", LightningModule=_LightningModule, Trainer=object)
stub("pytorch_lightning.callbacks", ModelCheckpoint=object)
stub("pytorch_lightning.logging", TestTubeLogger=object)
stub("kornia")
stub("kornia.losses", ssim=lambda *a, **k: None)
stub("imageio")
stub("utils.visualization", visualize_depth=lambda *a, **k: None)      # needs torchvision + cv2 + PIL
ds = stub("datasets", dataset_dict={})
ds.__path__ = []                                                          # a package, so that datasets.depth_utils resolves
spec = importlib.util.spec_from_file_location("datasets.depth_utils", os.path.join(REF, "datasets", "depth_utils.py"))
du = importlib.util.module_from_spec(spec)
spec.loader.exec_module(du)
sys.modules["datasets.depth_utils"] = du

# ---- the hot path: this package under the names the reference imports -------------------------------------------------
import nerf_pl_amd  # noqa: E402
nerf_pl_amd.install()
from nerf_pl_amd.models import nerf as our_nerf, rendering as our_rendering  # noqa: E402

import train  # noqa: E402  (the reference's train.py, unmodified)
import eval as ref_eval  # noqa: E402  (the reference's eval.py, unmodified)
from opt import get_opts  # noqa: E402,F401

sys.argv = ["train.py", "--N_importance", "64", "--img_wh", "400", "400", "--noise_std", "0", "--optimizer", "adam",
            "--lr", "5e-4", "--lr_scheduler", "steplr", "--decay_step", "2", "4", "8", "--decay_gamma", "0.5"]
hparams = get_opts()
system = train.NeRFSystem(hparams)
report = {
    "train.render_rays_is_ours": train.render_rays is our_rendering.render_rays,
    "eval.render_rays_is_ours": ref_eval.render_rays is our_rendering.render_rays,
    "train.Embedding_is_ours": train.Embedding is our_nerf.Embedding and train.NeRF is our_nerf.NeRF,
    "system.models_are_ours": all(isinstance(m, our_nerf.NeRF) for m in system.models) and len(system.models) == 2,
    "system.embeddings_are_ours": all(isinstance(e, our_nerf.Embedding) for e in system.embeddings),
    "embedding_channels": [system.embedding_xyz.out_channels, system.embedding_dir.out_channels],
    "state_dict_keys": len(system.state_dict()),
}
# the reference's forward calls render_rays POSITIONALLY with 10 arguments (train.py:55-64): our signature must take them
src = inspect.getsource(train.NeRFSystem.forward)
report["forward_calls_render_rays"] = "render_rays(self.models" in src.replace("\n", "").replace(" ", "").replace("\\", "")
sig = list(inspect.signature(our_rendering.render_rays).parameters)
report["render_rays_signature"] = sig
# the reference's own optimizer / scheduler factories accept our modules (utils/__init__.py:10-53)
opt = train.get_optimizer(hparams, system.models)
sched = train.get_scheduler(hparams, opt)
report["optimizer"] = type(opt).__name__
report["optimizer_params"] = sum(p.numel() for g in opt.param_groups for p in g["params"])
report["scheduler"] = type(sched).__name__
# eval.py's batched_inference keeps its signature and closes over `dataset.white_back` + our render_rays
report["eval_batched_inference_args"] = list(inspect.signature(ref_eval.batched_inference).parameters)
report["loss"] = type(system.loss).__name__
# checkpoints of the reference load by key (utils/__init__.py:55-76)
ck = {"state_dict": {"nerf_coarse." + k: v for k, v in our_nerf.NeRF().state_dict().items()}}
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "dropin_ckpt_%d.ckpt" % os.getpid())
torch.save(ck, path)
m = our_nerf.NeRF()
train.load_ckpt(m, path, model_name="nerf_coarse")
report["load_ckpt_roundtrip"] = all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), ck["state_dict"].values()))
os.remove(path)

# ---- with a GPU present: the reference's OWN training_step (train.py:103-117), forward chunk loop (train.py:49-71), loss
# (losses.py), optimizer (utils/__init__.py:10-30) and scheduler run a few real steps on top of this package's kernels, plain and
# under stock torch DistributedDataParallel at world size 1 (what Lightning's DDP wrapper does, train.py:174-175) ----
report["gpu"] = torch.cuda.is_available()
if report["gpu"]:
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    system = system.to(dev)
    (opt,), (sch,) = system.configure_optimizers()                      # the reference's method: sets system.optimizer
    g = torch.Generator().manual_seed(0)
    o = torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(512, 3, generator=g)
    d = torch.nn.functional.normalize(0.8 * torch.randn(512, 3, generator=g) - o, dim=-1)
    batch = {"rays": torch.cat([o, d, torch.full((512, 1), 2.0), torch.full((512, 1), 6.0)], 1).to(dev),
             "rgbs": torch.rand(512, 3, generator=g).to(dev)}
    losses = []
    for i in range(5):
        out = system.training_step(batch, i)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    report["training_step_losses"] = losses
    report["training_step_keys"] = sorted(out.keys())
    report["training_step_grads_finite"] = all(bool(torch.isfinite(p.grad).all()) for p in system.parameters() if p.grad is not None)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(system, device_ids=[0])
        res = ddp(batch["rays"])                                          # -> train.NeRFSystem.forward
        loss = system.loss(res, batch["rgbs"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        report["ddp_world1_loss"] = float(loss)
    finally:
        dist.destroy_process_group()
print("DROPIN_REPORT " + json.dumps(report))
