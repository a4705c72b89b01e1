#!/usr/bin/env python3
"""1x1 data-gradient launches (gate + residual + BatchNorm-reverse sums) at the bottleneck conv1 shapes: kernel variants timed interleaved."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops, resnet
B = 256
for name, H, Cin, Cout, nbn, k in (("l1 conv1 dgrad", 64, 64, 256, 1, 1), ("l1.1 (2 units)", 64, 64, 256, 2, 1), ("l2 conv1 dgrad", 32, 128, 512, 1, 1),
                                   ("l3 conv1 dgrad", 16, 256, 1024, 1, 1), ("l1 conv2 dgrad (3x3)", 64, 64, 64, 1, 3)):
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = resnet.pack_conv_weight(torch.randn(Cout, Cin, k, k) * 0.05, torch.bfloat16).cuda()
    mask, res = torch.randn(B, H, H, Cout, device="cuda").bfloat16(), torch.randn(B, H, H, Cout, device="cuda").bfloat16()
    bn = [(torch.randn(B, H, H, Cout, device="cuda").bfloat16(), torch.rand(2, Cout, device="cuda") + 0.5, ops.stat_unit(Cout, "cuda")) for _ in range(nbn)]
    nbytes = 2.0 * (x.numel() + (3 + nbn) * mask.numel())
    tiles = [8, 9] if k == 1 else [2, 8, 10]          # forced variants: 8 = 256x256 LDS-DMA, 9 / 10 = the streaming kernels, 2 = 128x128
    times = {t: [] for t in tiles}
    for r in range(6):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.conv2d_nhwc(x, w, k, k, 1, k // 2, residual=None if k == 3 else res, mask=mask, bn=bn, tile=t)
            e1.record(); torch.cuda.synchronize()
            if r: times[t].append(e0.elapsed_time(e1) * 200)
    print(name, {t: f"{statistics.median(v):.1f} us = {nbytes / statistics.median(v) / 1e6:.2f} TB/s" for t, v in times.items()}, flush=True)
