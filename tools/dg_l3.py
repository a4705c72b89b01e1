#!/usr/bin/env python3
"""a bottleneck's conv1 data gradient (python tools/dg_l3.py l1|l2|l3|l4; default layer3: 256 -> 1024 at 16 x 16; gate as bits + residual + one consumer's BatchNorm-reverse sums): kernel variants"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops, resnet
B = 256
H, Cin, Cout = {"l3": (16, 256, 1024), "l2": (32, 128, 512), "l1": (64, 64, 256), "l4": (8, 512, 2048)}[sys.argv[1] if len(sys.argv) > 1 else "l3"]
x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
w = resnet.pack_conv_weight(torch.randn(Cout, Cin, 1, 1) * 0.05, torch.bfloat16).cuda()
mask, res = torch.randn(B, H, H, Cout, device="cuda").bfloat16(), torch.randn(B, H, H, Cout, device="cuda").bfloat16()
bits = torch.zeros(B, H, H, Cout // 8, device="cuda", dtype=torch.uint8)
by = torch.randn(B, H, H, Cout, device="cuda").bfloat16()
for nbn, use_bits in ((1, True), (1, False), (0, True)):
    bn = [(by, torch.rand(2, Cout, device="cuda") + 0.5, ops.stat_unit(Cout, "cuda"))] if nbn else None
    nbytes = 2.0 * (x.numel() + (2 + nbn) * mask.numel()) + (bits.numel() if use_bits else 2.0 * mask.numel())
    times = {}
    for t in (0, 9, 8, 2, 3):          # launcher's choice, streaming kernel, phase-pipelined 256x256, register-staged 128x128, 256x256
        try:
            f = lambda: ops.conv2d_nhwc(x, w, 1, 1, 1, 0, residual=res, mask=mask, bn=bn, tile=t, mask_bits=bits if use_bits else None)
            f(); torch.cuda.synchronize()
        except Exception as e:
            times[t] = "n/a (" + str(e)[:40] + ")"; continue
        v = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            v.append(e0.elapsed_time(e1) * 200)
        times[t] = f"{statistics.median(v):.1f} us = {nbytes / statistics.median(v) / 1e6:.2f} TB/s"
    print(f"bn units {nbn}, gate bits {use_bits}:", times, flush=True)
