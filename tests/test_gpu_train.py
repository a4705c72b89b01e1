"""GPU parity of the train step's reverse kernels against torch autograd
run on the CPU oracle (the reference differentiates the same arithmetic with autograd,
hand/CrossModalHand.py:455-470).  fp32 tolerance 1e-4 relative to the tensor's scale."""
import numpy as np
import pytest
import torch

from conftest import load_golden, assert_close
from mhentropy_amd import synth, mano_pack

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _tables(seed=0):
    t = synth.mano_tables(seed)
    blob = _dev(mano_pack.pack_tables(t["shapedirs"], t["posedirs"], t["v_template"], t["J_regressor"], t["weights"],
                                      t["hands_components"][:45], t["hands_mean"]))
    from oracle import mano_ref
    return blob, mano_ref.tables_from_numpy(t)


@pytest.mark.parametrize("B,N,scale", [(3, 4, 0.3), (2, 5, 1.5)])
def test_mano_likelihood_backward_matches_autograd(gpu_lib, B, N, scale):
    """d sum_b g_b log_p_b / d (th45, det) with log_p_b the mean over the image's N hypotheses;
    scale=1.5 pushes th45/beta/th3 outside the soft priors' boxes so their gradients are exercised."""
    from mhentropy_amd import ops
    from oracle import network_ref
    blob, tb = _tables()
    rng = np.random.default_rng(5)
    R = N * B
    th45 = torch.as_tensor(rng.normal(0, scale, (R, 45)).astype(np.float32))
    det = torch.as_tensor(rng.normal(0, 1.0, (B, 16)).astype(np.float32))
    det[:, 3:13] *= 0.02 * scale
    det[:, 0:3] *= 2.0 * scale
    det[:, 13:] *= 0.2
    _, yn = synth.batch(3, B, with_image=False)
    y = {k: torch.as_tensor(v) for k, v in yn.items()}
    g = torch.as_tensor(rng.normal(0, 1, (B,)).astype(np.float32))
    th45_r, det_r = th45.clone().requires_grad_(True), det.clone().requires_grad_(True)
    z = network_ref.combine_z(det_r.repeat(N, 1), th45_r)
    lp = network_ref.forward_log_p(tb, z, y, N)["log_p"].reshape(N, B).mean(0)
    (lp * g).sum().backward()
    g45, gdet = ops.mano_joints_bwd(_dev(th45), _dev(det), blob, _dev(yn["crop_uv"]), _dev(yn["vis"]), _dev(g), N)
    assert_close(g45.cpu(), th45_r.grad, RTOL, what="d/d th45")
    assert_close(gdet.cpu(), det_r.grad, RTOL, what="d/d det")
