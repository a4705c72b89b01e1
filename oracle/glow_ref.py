"""ORACLE (test infrastructure, never shipped or measured as the product).  **PARITY UNPINNED.**

CPU restatement of `ConditionalGlow`, the flow the reference builds when `q_z_giv_i_model == 'glow'`
(/root/reference/hand/network.py:342-344: `ConditionalGlow(45, 512, 4, 2, context_features=512,
dropout_probability=0.2)`; call sites :693-694 `log_prob(z, context=feat) -> (log_prob, z)`, :736-742
`sample_and_log_prob(N, noise=noise, context=feat) -> (samples (B,N,D), log_prob (B,N), z)`).

The class lives in a third-party dependency that is ABSENT from /root/reference and not installed:
`git+https://github.com/nkolot/nflows.git`, unpinned (hand/environment.yml:284), the ProHMR fork of
bayesiains/nflows.  No reference test, fixture or golden vector exists for it, so nothing here could be checked
against the reference: this file restates the PUBLISHED nflows algorithm the fork builds on
(Durkan et al., nflows: `transforms.ActNorm`, `transforms.LULinear`, `transforms.AffineCouplingTransform`,
`nn.nets.ResidualNet` with GLU context gating, `distributions.StandardNormal`, `flows.Flow`), composed the way
Glow / ProHMR describe: per layer ActNorm -> LU-decomposed invertible linear -> affine coupling whose
scale/shift come from a context-conditioned residual MLP, with an alternating +-1 feature mask.
State-dict keys follow nflows' module tree (`_transform._transforms.{3l+0,1,2}...`).  Dropout: the identity (eval mode) unless
the caller supplies the masks of a train-mode pass (`masks=`, one [R, H] tensor per residual block in call order);
`use_batch_norm=False`.  Until the fork is reachable the HIP path is tested against THIS
restatement plus the flow's own identities (inverse(forward(x)) == x, log-det consistency).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def layer_prefix(l, j):
    return f"_transform._transforms.{3 * l + j}."


def lu_weight(sd, p, eps=1e-3):
    """nflows LULinear._create_lower_upper: unit-diagonal lower, softplus(+eps) diagonal upper"""
    D = sd[p + "unconstrained_upper_diag"].shape[0]
    dt = sd[p + "lower_entries"].dtype
    lower = torch.zeros(D, D, dtype=dt)
    li = np.tril_indices(D, k=-1)
    lower[li[0], li[1]] = sd[p + "lower_entries"]
    lower[range(D), range(D)] = 1.0
    upper = torch.zeros(D, D, dtype=dt)
    ui = np.triu_indices(D, k=1)
    upper[ui[0], ui[1]] = sd[p + "upper_entries"]
    diag = F.softplus(sd[p + "unconstrained_upper_diag"]) + eps
    upper[range(D), range(D)] = diag
    return lower @ upper, diag


def masks(features, num_layers):
    """nflows glow / SimpleRealNVP: mask = ones; mask[::2] = -1; flipped after every layer.
    Returns per layer (identity feature indices [mask <= 0], transform feature indices [mask > 0])."""
    m = torch.ones(features)
    m[::2] = -1
    out = []
    for _ in range(num_layers):
        idx = torch.arange(features)
        out.append((idx[m <= 0], idx[m > 0]))
        m = -m
    return out


def residual_net(sd, p, inputs, context, num_blocks, masks=None):
    """nflows nn.nets.ResidualNet with context (no batch norm).  masks: None = eval mode (dropout is the identity); an iterator of [R, H]
    tensors holding 0 or 1 / (1 - p) = train mode with THESE dropout masks, one per residual block in call order (nn.Dropout's own
    draws cannot be matched across generators: the tests hand over the masks the device drew)"""
    t = F.linear(torch.cat([inputs, context], 1), sd[p + "initial_layer.weight"], sd[p + "initial_layer.bias"])
    for b in range(num_blocks):
        q = p + f"blocks.{b}."
        u = F.relu(t)
        u = F.linear(u, sd[q + "linear_layers.0.weight"], sd[q + "linear_layers.0.bias"])
        u = F.relu(u)
        if masks is not None:
            u = u * next(masks)                       # nflows ResidualBlock: activation -> dropout -> linear_layers[1]
        u = F.linear(u, sd[q + "linear_layers.1.weight"], sd[q + "linear_layers.1.bias"])
        g = F.linear(context, sd[q + "context_layer.weight"], sd[q + "context_layer.bias"])
        t = t + u * torch.sigmoid(g)                  # F.glu(cat(u, g)) = u * sigmoid(g)
    return F.linear(t, sd[p + "final_layer.weight"], sd[p + "final_layer.bias"])


def _scale_shift(params, T):
    """nflows AffineCouplingTransform._scale_and_shift: shift = first half, scale = sigmoid(second half + 2) + 1e-3"""
    return torch.sigmoid(params[:, T:] + 2.0) + 1e-3, params[:, :T]


def transform_forward(sd, x, context, num_layers, num_blocks, drop=None):
    """data -> noise, with log|det|"""
    D = x.shape[1]
    logdet = x.new_zeros(x.shape[0])
    for l, (idf, trf) in enumerate(masks(D, num_layers)):
        p = layer_prefix(l, 0)
        x = torch.exp(sd[p + "log_scale"]) * x + sd[p + "shift"]
        logdet = logdet + sd[p + "log_scale"].sum()
        p = layer_prefix(l, 1)
        W, diag = lu_weight(sd, p)
        x = F.linear(x, W, sd[p + "bias"])
        logdet = logdet + torch.log(diag).sum()
        p = layer_prefix(l, 2)
        params = residual_net(sd, p + "transform_net.", x[:, idf], context, num_blocks, drop)
        scale, shift = _scale_shift(params, trf.numel())
        y = x.clone()
        y[:, trf] = x[:, trf] * scale + shift
        logdet = logdet + torch.log(scale).sum(1)
        x = y
    return x, logdet


def transform_inverse(sd, z, context, num_layers, num_blocks, drop=None):
    """noise -> data, with log|det| of the inverse"""
    D = z.shape[1]
    logdet = z.new_zeros(z.shape[0])
    ms = masks(D, num_layers)
    for l in reversed(range(num_layers)):
        idf, trf = ms[l]
        p = layer_prefix(l, 2)
        params = residual_net(sd, p + "transform_net.", z[:, idf], context, num_blocks, drop)
        scale, shift = _scale_shift(params, trf.numel())
        y = z.clone()
        y[:, trf] = (z[:, trf] - shift) / scale
        logdet = logdet - torch.log(scale).sum(1)
        p = layer_prefix(l, 1)
        W, diag = lu_weight(sd, p)
        y = torch.linalg.solve(W.double(), (y - sd[p + "bias"]).double().t()).t().to(z.dtype)
        logdet = logdet - torch.log(diag).sum()
        p = layer_prefix(l, 0)
        z = (y - sd[p + "shift"]) / torch.exp(sd[p + "log_scale"])
        logdet = logdet - sd[p + "log_scale"].sum()
    return z, logdet


def std_normal_log_prob(z):
    return -0.5 * (z * z).sum(1) - 0.5 * z.shape[1] * math.log(2 * math.pi)


def log_prob(sd, x, context, num_layers=4, num_blocks=2, masks=None):
    """Flow.log_prob as the fork returns it: (log_prob (R,), noise (R,D))"""
    z, logdet = transform_forward(sd, x, context, num_layers, num_blocks, None if masks is None else iter(masks))
    return std_normal_log_prob(z) + logdet, z


def sample_and_log_prob(sd, noise, context, num_layers=4, num_blocks=2, masks=None):
    """noise (B,N,D), context (B,F) -> samples (B,N,D), log_prob (B,N), noise: rows are batch-major
    (nflows repeats each context row N times, `torchutils.repeat_rows`)."""
    B, N, D = noise.shape
    z = noise.reshape(B * N, D)
    ctx = context.repeat_interleave(N, 0)
    x, logdet = transform_inverse(sd, z, ctx, num_layers, num_blocks, None if masks is None else iter(masks))
    lp = std_normal_log_prob(z) - logdet
    return x.reshape(B, N, D), lp.reshape(B, N), noise
