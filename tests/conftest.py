import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_PATH = os.path.join(ROOT, "tests", "golden", "reference_golden.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    """every `gpu`-marked test is skipped on a box without a GPU, whether or not it takes the `dev` fixture"""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no GPU visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    z = np.load(GOLDEN_PATH, allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" else a
    return out


@pytest.fixture(scope="session")
def golden_grads():
    """gr3 (BASELINE configs[2] shape): all 48 gradient tensors of the reference's training loss, in full (oracle/make_golden.py)"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_golden_grads.npz"), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
