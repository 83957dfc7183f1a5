import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(REPO, "voltrix-spmm_amd")
for p in (REPO, PKG_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# JIT cache in-tree: kernels built by __graft_entry__.build() here travel to the GPU box with the snapshot
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(PKG_ROOT, ".jit_cache"))

GOLDEN = os.path.join(REPO, "tests", "golden")
CSR_FIXTURES = ("toy40", "cora_like", "sprandom_2708", "skewed_1005")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_csr_fixture(name):
    g = np.load(os.path.join(GOLDEN, f"csr_{name}.npz"))
    return {k: g[k] for k in g.files}


@pytest.fixture(params=CSR_FIXTURES)
def csr_fixture(request):
    return load_csr_fixture(request.param)


@pytest.fixture(scope="session")
def cuda_device():
    """GPU tests must fail loudly (never skip) when the device or the HIP extension is missing."""
    import torch

    assert torch.cuda.is_available(), "GPU test selected (-m gpu) but no GPU is visible"
    return torch.device("cuda:0")
