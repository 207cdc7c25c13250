"""Drop-in proof: the reference's REAL train.py / eval.py import and construct on top of this package after
`nerf_pl_amd.install()` (skipped where /root/reference is absent, i.e. on the GPU box).  Runs in a subprocess because the
reference's top-level module names (`utils`, `datasets`, `metrics`, `eval`, ...) would shadow installed packages."""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "train.py")), reason="reference tree not present")
def test_reference_train_and_eval_run_on_this_package():
    r = subprocess.run([sys.executable, os.path.join(HERE, "_dropin_probe.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DROPIN_REPORT ")]
    assert line, r.stdout[-2000:]
    rep = json.loads(line[-1][len("DROPIN_REPORT "):])
    for k in ("train.render_rays_is_ours", "eval.render_rays_is_ours", "train.Embedding_is_ours", "system.models_are_ours",
              "system.embeddings_are_ours", "forward_calls_render_rays", "load_ckpt_roundtrip"):
        assert rep[k] is True, (k, rep)
    assert rep["embedding_channels"] == [63, 27]
    assert rep["state_dict_keys"] == 48                       # 2 models x 12 layers x (weight, bias)
    # train.py:55-64 passes these ten positionally
    assert rep["render_rays_signature"][:10] == ["models", "embeddings", "rays", "N_samples", "use_disp", "perturb", "noise_std",
                                                  "N_importance", "chunk", "white_back"]
    assert rep["render_rays_signature"][10] == "test_time"    # eval.py:69-79 passes it by keyword
    assert rep["optimizer"] == "Adam" and rep["optimizer_params"] == 2 * 595844 and rep["scheduler"] == "MultiStepLR"
    assert rep["eval_batched_inference_args"] == ["models", "embeddings", "rays", "N_samples", "N_importance", "use_disp", "chunk",
                                                  "white_back"]
    assert rep["loss"] == "MSELoss"
    if rep.get("gpu"):         # a box with BOTH the reference tree and an MI355X: the reference's own training_step ran on our kernels
        ls = rep["training_step_losses"]
        assert rep["training_step_keys"] == ["log", "loss", "progress_bar"] and rep["training_step_grads_finite"]
        assert all(l == l and l < 10 for l in ls) and min(ls[2:]) < ls[0], ls
        assert rep["ddp_world1_loss"] == rep["ddp_world1_loss"]
