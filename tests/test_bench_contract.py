"""bench.py's launch contract: `--gpus N` either runs N ranks or fails loudly — it never prints an
`n_gpus: 1` line for a request of N > 1.  The CPU half needs no GPU: the refusals happen before any device work."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def test_world_size_mismatch_is_refused_before_any_work():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and r.stdout.strip() == "" and "WORLD_SIZE=4 but --gpus 2" in r.stderr


@pytest.mark.skipif(__import__("torch").cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_more_gpus_than_visible_fails_loudly_without_a_json_line():
    r = _run(["--gpus", "2"])
    assert r.returncode == 3 and r.stdout.strip() == "" and "only" in r.stderr


@pytest.mark.gpu
def test_torchrun_world1_rccl_path_agrees_with_the_plain_run():
    """The launcher path the driver uses for N > 1, at world size 1 with the gradients routed through RCCL (GradSync over a
    one-rank communicator): same step, same line shape, ms_per_step within 10 % of the plain run."""
    common = ["--gpus", "1", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-extras"]
    plain = _run(common)
    assert plain.returncode == 0, plain.stderr[-2000:]
    a = json.loads(plain.stdout.strip().splitlines()[-1])
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", BENCH] + common + ["--force-dist"], capture_output=True, text=True, env=env, timeout=600,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads(r.stdout.strip().splitlines()[-1])
    print("plain %.4f ms/step, torchrun + RCCL world-1 %.4f ms/step (%s)" % (a["ms_per_step"], b["ms_per_step"], b["config"]["grad_sync"]))
    assert a["n_gpus"] == b["n_gpus"] == 1 and b["config"]["rccl_nranks"] == 1 and a["config"]["rccl_nranks"] is None
    assert a["dtype"] == b["dtype"] and a["config"]["workload"] == b["config"]["workload"]
    assert abs(b["ms_per_step"] - a["ms_per_step"]) <= 0.10 * a["ms_per_step"], (a["ms_per_step"], b["ms_per_step"])
