import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU; opt in with ASIS_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("ASIS_SLOW"):
        return
    skip = pytest.mark.skip(reason="slow CPU case: set ASIS_SLOW=1")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def load_golden(name: str):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), map_location="cpu")


def golden_err(t: torch.Tensor, g: dict) -> float:
    """rel-L2 of tensor ``t`` against a sub-sampled golden entry (see tests/golden/make_golden.py:sub)."""
    assert list(t.shape) == g["shape"].tolist(), (list(t.shape), g["shape"].tolist())
    flat = t.detach().float().cpu().reshape(-1)[:: int(g["step"])]
    return rel_l2(flat, g["vals"])


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
