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


def free_port():
    """a TCP port that is free right now on 127.0.0.1 (bind to port 0, read it back): a fixed rendezvous port fails with EADDRINUSE when a
    concurrent session or a socket in TIME_WAIT holds it - and looks like a GPU / RCCL failure"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


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


def check_ho3d_against_fixture(img, tgt, g, what, img_tol=0, tol=1e-6):
    """one sample of the HO3D pipeline (image [3,256,256] f32 in [-1,1], target dict of arrays) against a reference-generated fixture
    tests/golden/ho3d_*.npz: pixels / masks bit-exact (integer work), floating-point targets to `tol` of their scale"""
    u8 = np.rint((np.asarray(img, np.float64) * 0.5 + 0.5) * 255).astype(np.int64)
    assert np.abs(u8[:, ::4, ::4] - g["image_u8_sub"]).max() <= img_tol, what + ": image pixels"
    assert np.array_equal(u8.sum((1, 2)), g["image_u8_sum"]) or img_tol, what + ": image checksum"
    for k in ("hand_mask", "object_mask"):
        assert np.array_equal(np.packbits(np.asarray(tgt[k]).astype(bool)), g[k]), what + ": " + k
    d = np.asarray(tgt["depth"], np.float32)
    assert_close(d[::4, ::4], g["depth_sub"], 1e-6, what=what + ": depth")
    assert abs(float(d.astype(np.float64).sum()) - float(g["depth_sum"])) <= 1e-5 * abs(float(g["depth_sum"])) + 1e-6, what + ": depth checksum"
    assert np.array_equal(np.asarray(tgt["vis"]).reshape(-1), g["vis"].reshape(-1)), what + ": vis"
    for k in ("crop_uv", "original_pose3d", "pose3d", "st", "scale", "crop_center", "crop_size", "pose3d_root", "rot_mat_inv", "_rot_mat", "uvd"):
        assert_close(np.asarray(tgt[k]).reshape(g[k].shape), g[k], tol, 1e-7, what=what + ": " + k)
    v = np.asarray(tgt["verts"], np.float64).sum()
    assert abs(v - float(g["verts_sum"])) <= 1e-5 * abs(float(g["verts_sum"])) + 1e-3, what + ": verts checksum"
