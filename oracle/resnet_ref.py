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


def _q(t):
    """round to bf16 and back: a storage point of the bf16 performance mode"""
    return t.bfloat16().float()


def forward_bf16_storage(sd, x, arch="resnet50", training=True, taps=None):
    """The same network with every stored tensor (input, weights, raw conv outputs, activations) rounded
    to bf16 and all arithmetic in f32 - the rounding points of the product's bf16 encoder mode
    (mhentropy_amd/resnet.py): batch statistics of the conv output AS STORED (bf16-rounded - the tensor the
    normalisation is applied to; the first version of the kernels summed the f32 accumulators, which differs by the mean of
    the rounding errors), BN+ReLU evaluated in f32 on the bf16-stored conv output and stored as bf16."""
    kind, blocks, _ = CFG[arch]

    def conv_bn(inp, cname, bname, stride=1, pad=0, relu=True, res=None):
        y = _q(F.conv2d(inp, _q(sd[cname + ".weight"]), None, stride, pad))
        if training:
            mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
        else:
            mean, var = sd[bname + ".running_mean"], sd[bname + ".running_var"]
        sc = sd[bname + ".weight"] / torch.sqrt(var + BN_EPS)
        sh = sd[bname + ".bias"] - mean * sc
        return y, sc.view(1, -1, 1, 1), sh.view(1, -1, 1, 1)

    y, sc, sh = conv_bn(_q(x), "conv1", "bn1", 2, 3)
    a = _q(F.max_pool2d(F.relu(y * sc + sh), 3, 2, 1))
    for li, nb in enumerate(blocks):
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 0) else 1
            p = f"layer{li + 1}.{bi}"
            if kind == "bottleneck":
                y, sc, sh = conv_bn(a, p + ".conv1", p + ".bn1")
                y, sc, sh = conv_bn(_q(F.relu(y * sc + sh)), p + ".conv2", p + ".bn2", stride, 1)
                y, sc, sh = conv_bn(_q(F.relu(y * sc + sh)), p + ".conv3", p + ".bn3")
            else:
                y, sc, sh = conv_bn(a, p + ".conv1", p + ".bn1", stride, 1)
                y, sc, sh = conv_bn(_q(F.relu(y * sc + sh)), p + ".conv2", p + ".bn2", 1, 1)
            if (p + ".downsample.0.weight") in sd:
                yd, scd, shd = conv_bn(a, p + ".downsample.0", p + ".downsample.1", stride)
                idt = yd * scd + shd
            else:
                idt = a
            a = _q(F.relu(y * sc + sh + idt))
            if taps is not None:
                taps[p] = a
    return torch.flatten(F.adaptive_avg_pool2d(a, 1), 1)


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
