#!/usr/bin/env python3
"""Per-layer time of the weight-gradient kernel over ResNet-50's conv shapes at B=256 (bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops
B = int(os.environ.get("B", 256))
ops.WGRAD_SLABS = os.environ.get("SLABS", "1") == "1"
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
shapes = [(64, 64, 1, 1, 64), (64, 64, 3, 1, 64), (64, 256, 1, 1, 64), (256, 64, 1, 1, 64), (256, 128, 1, 1, 64), (128, 128, 3, 2, 64),
          (128, 512, 1, 1, 32), (512, 128, 1, 1, 32), (128, 128, 3, 1, 32), (512, 256, 1, 1, 32), (256, 256, 3, 2, 32), (256, 1024, 1, 1, 16),
          (1024, 256, 1, 1, 16), (256, 256, 3, 1, 16), (1024, 512, 1, 1, 16), (512, 512, 3, 2, 16), (512, 2048, 1, 1, 8), (2048, 512, 1, 1, 8),
          (512, 512, 3, 1, 8), (512, 512, 1, 1, 1)]
tot = 0
for cin, cout, k, s, H in shapes:
    if H == 1:
        x = torch.randn(16384, 1, 1, cin, device="cuda").to(dt); Ho = 1
    else:
        x = torch.randn(B, H, H, cin, device="cuda").to(dt); Ho = (H + 2 * (k // 2) - k) // s + 1
    gy = torch.randn(x.shape[0], Ho, Ho, cout, device="cuda").to(dt)
    dw = torch.zeros(cout, k * k * cin, device="cuda")
    for _ in range(2): ops.conv_wgrad(x, gy, k, k, s, k // 2, dw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv_wgrad(x, gy, k, k, s, k // 2, dw)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    fl = 2.0 * x.shape[0] * Ho * Ho * cout * k * k * cin
    tot += t
    print(f"cin {cin:5d} cout {cout:5d} k{k} s{s} H{H:3d}: {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TF")
print("sum ms", tot)
