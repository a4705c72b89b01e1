"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the reference's train step (/root/reference/hand/CrossModalHand.py:455-470 with the
optimizer of :191-203): total_loss = mean(-log_p) (hand/criteria.py:55,173), autograd backward,
clip_grad_norm_(parameters, 1.0), torch.optim.Adam(lr) step - on the oracle's functional get_loss
(oracle/network_ref.py), so the gradients are torch autograd's on the reference arithmetic.
Pinned by the reference gradients in tests/golden/mhent_*.npz (feat, det_head, two flow layers).
"""
import torch

from . import network_ref


def _is_param(k):
    return not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked") or k.endswith("mask"))


def loss_and_grads(sd, tb, x, y, z0, N, arch="resnet50", bf16_storage=False):
    """sd: flat state dict of float tensors (reference key names).  Returns (get_loss dict, total, {name: grad}).
    bf16_storage: the trunk's stored tensors (and the gradients passing them) rounded to bf16, network_ref.encoder."""
    p = {k: (v.clone().requires_grad_(True) if _is_param(k) and v.is_floating_point() else v.clone()) for k, v in sd.items()}
    out = network_ref.get_loss(p, tb, x, y, z0, N, arch=arch, training=True, bf16_storage=bf16_storage)
    total = (-out["log_p"]).mean()
    total.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items() if isinstance(v, torch.Tensor) and v.requires_grad}
    buffers = {k: v for k, v in p.items() if not (isinstance(v, torch.Tensor) and v.requires_grad)}
    return out, total.detach(), grads, buffers


def clip_and_adam(params, grads, state, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (defaults) on dicts of tensors; state = {'t', 'm', 'v'}"""
    total_norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    out = {}
    for k, p in params.items():
        g = grads[k] * coef
        m = state.setdefault("m", {}).get(k, torch.zeros_like(p))
        v = state.setdefault("v", {}).get(k, torch.zeros_like(p))
        m = betas[0] * m + (1 - betas[0]) * g
        v = betas[1] * v + (1 - betas[1]) * g * g
        state["m"][k], state["v"][k] = m, v
        out[k] = p - lr / (1 - betas[0] ** t) * m / (v.sqrt() / (1 - betas[1] ** t) ** 0.5 + eps)
    return out, total_norm
