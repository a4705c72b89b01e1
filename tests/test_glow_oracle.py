"""CPU: the Glow restatement's own identities (it has no reference fixture to be pinned by - PARITY UNPINNED,
oracle/glow_ref.py header): inverse(forward(x)) == x, densities agree from both directions, LU determinant."""
import numpy as np
import torch

from mhentropy_amd import synth
from oracle import glow_ref


def test_glow_restatement_is_a_bijection_with_consistent_density():
    sd = {k: torch.as_tensor(v).double() for k, v in synth.glow_state(0, 45, 64, 4, 2, 32).items()}
    rng = np.random.default_rng(0)
    B, N = 3, 4
    noise = torch.as_tensor(rng.normal(0, 1, (B, N, 45)))
    ctx = torch.as_tensor(rng.normal(0, 1, (B, 32)))
    x, lp, _ = glow_ref.sample_and_log_prob(sd, noise, ctx)
    lq, z = glow_ref.log_prob(sd, x.reshape(B * N, 45), ctx.repeat_interleave(N, 0))
    assert (z - noise.reshape(B * N, 45)).abs().max() < 1e-9
    assert (lq - lp.reshape(-1)).abs().max() < 1e-9
    W, diag = glow_ref.lu_weight(sd, glow_ref.layer_prefix(1, 1))
    assert abs(torch.linalg.slogdet(W)[1] - torch.log(diag).sum()) < 1e-10
    ms = glow_ref.masks(45, 4)
    assert ms[0][0].tolist() == list(range(0, 45, 2)) and ms[1][0].tolist() == list(range(1, 45, 2))     # alternating +-1 mask
