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
print("DROPIN_REPORT " + json.dumps(report))
