#!/usr/bin/env python3
"""CPU experiment (oracle only, no GPU): how far does a bf16-STORAGE ResNet-50 trunk move the gradient of mean(-log_p) from the float64
gradient, with exact (f32) arithmetic everywhere else?  oracle/resnet_ref.forward_bf16_storage rounds every stored tensor of the forward
pass to bf16 (weights, raw conv outputs, activations) and - autograd of t.bfloat16().float() - every gradient that passes a storage point.
Variants: both roundings (what a bf16 train step does by construction), forward only (straight-through gradient), backward only.
Yardstick for tests/test_gpu_train_parity.py: the HIP bf16 step has the same rounding points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mhentropy_amd import synth
from oracle import network_ref, resnet_ref, mano_ref

B, N, S = int(os.environ.get("B", 32)), int(os.environ.get("N", 64)), int(os.environ.get("S", 256))
MODES = os.environ.get("MODES", "both,fwd,bwd").split(",")


class QF(torch.autograd.Function):          # round forward, pass the gradient through
    @staticmethod
    def forward(ctx, t):
        return t.bfloat16().float()
    @staticmethod
    def backward(ctx, g):
        return g


class QB(torch.autograd.Function):          # pass forward, round the gradient
    @staticmethod
    def forward(ctx, t):
        return t.clone()
    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


sd = {}
sd.update({"feat_extractor.res." + k: v for k, v in synth.resnet_state(50, "resnet50").items()})
sd.update(synth.head_state(51, 2048))
sd.update({"q_z_giv_i." + k: v for k, v in synth.flow_state(52, 45, 512, (512, 512), 6).items()})
sd = {k: torch.as_tensor(v) for k, v in sd.items()}
GAIN = float(os.environ.get("RES_GAIN", "1"))           # < 1: residual branches damped (bn3 gamma scaled): a near-identity, smoother trunk
for k in sd:
    if k.endswith("bn3.weight"):
        sd[k] = sd[k] * GAIN
xn, yn = synth.batch(53, B, image_size=S)
if os.environ.get("IMG") == "structured":
    xn = synth.structured_images(53, B, S)
z0 = torch.as_tensor(synth.noise(53, N * B))
x, y = torch.as_tensor(xn), {k: torch.as_tensor(v) for k, v in yn.items()}
isp = lambda k: not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked") or k.endswith("mask"))


def run(DT, trunk):
    tb = mano_ref.tables_from_numpy(synth.mano_tables(0), dtype=DT)
    p = {k: (v.to(DT).clone().requires_grad_(True) if isp(k) and v.is_floating_point() else (v.to(DT) if v.is_floating_point() else v)) for k, v in sd.items()}
    yy = {k: (v.to(DT) if v.is_floating_point() else v) for k, v in y.items()}
    f = trunk(network_ref.sub(p, "feat_extractor.res."), x.to(DT))
    f.retain_grad()
    feat = F.linear(f, p["feat_extractor.l1.0.weight"], p["feat_extractor.l1.0.bias"])
    out = network_ref.reverse_kld(p, tb, feat, yy, z0.to(DT), N)
    total = (-out["log_p"]).mean()
    total.backward()
    g = {k: v.grad.double() for k, v in p.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
    g["_g_f"] = f.grad.double()
    return float(total), g


GROUPS = [("stem", ("feat_extractor.res.conv1.", "feat_extractor.res.bn1.")), ("layer1", ("feat_extractor.res.layer1.",)),
          ("layer2", ("feat_extractor.res.layer2.",)), ("layer3", ("feat_extractor.res.layer3.",)), ("layer4", ("feat_extractor.res.layer4.",)),
          ("l1", ("feat_extractor.l1.",)), ("flow", ("q_z_giv_i.",)), ("g_f", ("_g_f",))]


def report(tag, g, ref):
    for name, pre in GROUPS:
        ks = [k for k in ref if k.startswith(pre) and k in g]
        a = torch.cat([g[k].reshape(-1) for k in ks]); b = torch.cat([ref[k].reshape(-1) for k in ks])
        print(f"  {tag:10s} {name:7s} rel-L2 {float((a - b).norm() / b.norm()):.3e}  cosine {float((a * b).sum() / (a.norm() * b.norm())):.5f}", flush=True)


t = time.time()
l64, g64 = run(torch.float64, lambda s, xx: resnet_ref.forward(s, xx, "resnet50", True))
print(f"f64 oracle: total {l64:.6f} ({time.time() - t:.0f} s)", flush=True)
l32, g32 = run(torch.float32, lambda s, xx: resnet_ref.forward(s, xx, "resnet50", True))
print(f"f32 oracle: total {l32:.6f}")
report("f32", g32, g64)
for mode in MODES:
    q = {"both": resnet_ref._q, "fwd": QF.apply, "bwd": QB.apply}[mode]
    old = resnet_ref._q
    resnet_ref._q = q
    try:
        lb, gb = run(torch.float32, lambda s, xx: resnet_ref.forward_bf16_storage(s, xx, "resnet50", True))
    finally:
        resnet_ref._q = old
    print(f"bf16 storage ({mode}): total {lb:.6f}")
    report("bf16-" + mode, gb, g64)
