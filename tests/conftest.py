import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from safetensors.torch import load_file

    def load(name):
        return load_file(os.path.join(GOLDEN, name))
    return load


def seeded(shape, seed, dtype=None, scale=1.0):
    import torch
    g = torch.Generator("cpu").manual_seed(seed)
    x = torch.randn(shape, generator=g, dtype=torch.float32) * scale
    return x.to(torch.bfloat16 if dtype is None else dtype)
