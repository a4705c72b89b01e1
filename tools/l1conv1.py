#!/usr/bin/env python3
"""layer1.0's conv1 (1x1, 64 -> 64 on the pooled stem output, bn1 + ReLU on its load): kernel variants"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops, resnet
B, H, Cin, Cout = 256, 64, 64, 64
x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
w = resnet.pack_conv_weight(torch.randn(Cout, Cin, 1, 1) * 0.1, torch.bfloat16).cuda()
sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
st = ops.stat_unit(Cout, "cuda")
for bn in (True, False):
    out = {}
    for t in (0, 1, 2, 9):
        kw = dict(in_scale=sc, in_shift=sh, relu_in=True) if bn else {}
        try:
            f = lambda: ops.conv2d_nhwc(x, w, 1, 1, 1, 0, stats=st, tile=t, **kw)
            f(); torch.cuda.synchronize()
        except Exception as e:
            out[t] = "n/a"; continue
        v = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            v.append(e0.elapsed_time(e1) * 200)
        out[t] = f"{statistics.median(v):.1f} us"
    print("bn on load" if bn else "plain", out, "tile choice", ops.conv_tile_choice(B, H, H, Cin, Cout, 1, 1, 0, torch.bfloat16, 1 if bn else 0))
