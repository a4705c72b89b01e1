import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through the C-ABI HIP library)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def assert_close(a, b, rtol, atol=0.0, what=""):
    """|a-b| <= atol + rtol*max|b| elementwise (relative to the tensor's scale)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    tol = atol + rtol * (np.abs(b).max() if b.size else 0.0)
    err = np.abs(a - b).max() if b.size else 0.0
    assert err <= tol, f"{what}: max|diff|={err:.3e} > tol={tol:.3e}"


@pytest.fixture(scope="session")
def gpu_lib():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mhentropy_amd import _lib
    return _lib.lib()
