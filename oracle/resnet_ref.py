"""ORACLE (test infrastructure, never shipped or measured as the product).

The reference's image encoder is torchvision 0.9.0 `resnet18/50` with
`fc = Identity` (/root/reference/hand/network.py:54-61, environment.yml:227).
torchvision is a third-party dependency that is NOT under /root/reference and is
not installed here, so this file restates the published ResNet v1.5 topology
(He et al. 2015; stride on the 3x3 conv of a bottleneck) on torch primitives
(conv2d / batch_norm / max_pool2d / adaptive_avg_pool2d), using torchvision's
state_dict key names.  PARITY UNPINNED with respect to torchvision itself (no
reference test or fixture exists for it); pinned with respect to the torch
primitives it is written in.
"""
import torch
import torch.nn.functional as F

CFG = {
    "resnet18": ("basic", (2, 2, 2, 2), 512),
    "resnet50": ("bottleneck", (3, 4, 6, 3), 2048),
}
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _bn(sd, name, x, training, stats_out=None):
    rm, rv = sd[name + ".running_mean"], sd[name + ".running_var"]
    if training:
        # functional batch_norm would update rm/rv in place; keep the oracle pure
        y = F.batch_norm(x, None, None, sd[name + ".weight"], sd[name + ".bias"], True, BN_MOMENTUM, BN_EPS)
        if stats_out is not None:
            n = x.numel() / x.shape[1]
            mean = x.mean((0, 2, 3))
            var = x.var((0, 2, 3), unbiased=False)
            stats_out[name] = (mean, var,
                               (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean,
                               (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * n / (n - 1))
        return y
    return F.batch_norm(x, rm, rv, sd[name + ".weight"], sd[name + ".bias"], False, BN_MOMENTUM, BN_EPS)


def forward(sd, x, arch="resnet50", training=True, stats_out=None, taps=None):
    """x (B,3,H,W) -> pooled feature (B,512|2048).  `taps` (dict) collects named
    intermediate activations for layer-by-layer parity tests."""
    kind, blocks, _ = CFG[arch]
    tap = (lambda k, v: taps.__setitem__(k, v)) if taps is not None else (lambda k, v: None)
    x = F.conv2d(x, sd["conv1.weight"], None, stride=2, padding=3)
    tap("conv1", x)
    x = F.relu(_bn(sd, "bn1", x, training, stats_out))
    x = F.max_pool2d(x, 3, 2, 1)
    tap("pool", x)
    for li, nb in enumerate(blocks):
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 0) else 1
            p = f"layer{li + 1}.{bi}"
            idt = x
            if kind == "bottleneck":
                o = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"]), training, stats_out))
                o = F.relu(_bn(sd, p + ".bn2", F.conv2d(o, sd[p + ".conv2.weight"], None, stride, 1), training, stats_out))
                o = _bn(sd, p + ".bn3", F.conv2d(o, sd[p + ".conv3.weight"]), training, stats_out)
            else:
                o = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), training, stats_out))
                o = _bn(sd, p + ".bn2", F.conv2d(o, sd[p + ".conv2.weight"], None, 1, 1), training, stats_out)
            if (p + ".downsample.0.weight") in sd:
                idt = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), training, stats_out)
            x = F.relu(o + idt)
            tap(p, x)
    return torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
