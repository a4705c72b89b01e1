#!/usr/bin/env python3
"""Per-shape micro-benchmark of the conv kernel over the distinct ResNet-50 layer shapes at the bench
batch (B=256, 256x256 input): time, TFLOP/s and algorithmic HBM GB/s (input + output once) per shape."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops, resnet

# (name, H, Cin, Cout, k, stride, count, fused_input_affine)
SHAPES = [
    ("stem 7x7/2 3->64", 256, 3, 64, 7, 2, 1, False),
    ("l1 1x1 64->64", 64, 64, 64, 1, 1, 1, False),
    ("l1 1x1 256->64", 64, 256, 64, 1, 1, 2, False),
    ("l1 3x3 64->64", 64, 64, 64, 3, 1, 3, True),
    ("l1 1x1 64->256", 64, 64, 256, 1, 1, 4, True),
    ("l2 1x1 256->128", 64, 256, 128, 1, 1, 1, False),
    ("l2 3x3/2 128->128", 64, 128, 128, 3, 2, 1, True),
    ("l2 1x1 128->512", 32, 128, 512, 1, 1, 4, True),
    ("l2 ds/2 256->512", 64, 256, 512, 1, 2, 1, False),
    ("l2 1x1 512->128", 32, 512, 128, 1, 1, 3, False),
    ("l2 3x3 128->128", 32, 128, 128, 3, 1, 3, True),
    ("l3 1x1 512->256", 32, 512, 256, 1, 1, 1, False),
    ("l3 3x3/2 256->256", 32, 256, 256, 3, 2, 1, True),
    ("l3 1x1 256->1024", 16, 256, 1024, 1, 1, 6, True),
    ("l3 ds/2 512->1024", 32, 512, 1024, 1, 2, 1, False),
    ("l3 1x1 1024->256", 16, 1024, 256, 1, 1, 5, False),
    ("l3 3x3 256->256", 16, 256, 256, 3, 1, 5, True),
    ("l4 1x1 1024->512", 16, 1024, 512, 1, 1, 1, False),
    ("l4 3x3/2 512->512", 16, 512, 512, 3, 2, 1, True),
    ("l4 1x1 512->2048", 8, 512, 2048, 1, 1, 3, True),
    ("l4 ds/2 1024->2048", 16, 1024, 2048, 1, 2, 1, False),
    ("l4 1x1 2048->512", 8, 2048, 512, 1, 1, 2, False),
    ("l4 3x3 512->512", 8, 512, 512, 3, 1, 2, True),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--no-affine", action="store_true")
    ap.add_argument("--tile", type=int, default=0, help="force kernel variant tile-1 (mhe_conv_desc.tile)")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    es = 2 if dt == torch.bfloat16 else 4
    B = args.batch
    tot_t = tot_ideal = 0.0
    print(f"{'shape':22s} {'n':>2s} {'us':>9s} {'TF/s':>8s} {'GB/s':>8s} {'hbm-ideal us':>12s} {'mfma-ideal us':>13s}")
    for name, H, Cin, Cout, k, stride, count, fused in SHAPES:
        if args.only and args.only not in name:
            continue
        ce = 16 // es
        Cp = (Cin + ce - 1) // ce * ce
        x = torch.randn(B, H, H, Cp, device="cuda").to(dt)
        w = resnet.pack_conv_weight(torch.randn(Cout, Cin, k, k) * 0.05, dt, Cp).cuda()
        pad = k // 2
        Ho = (H + 2 * pad - k) // stride + 1
        y = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=dt)
        st = None if args.no_stats else ops.stat_unit(Cout, "cuda")
        isc = ish = None
        if fused and not args.no_affine:
            isc, ish = torch.rand(Cp, device="cuda") + 0.5, torch.randn(Cp, device="cuda") * 0.1
        run = lambda: ops.conv2d_nhwc(x, w, k, k, stride, pad, in_scale=isc, in_shift=ish, relu_in=isc is not None,
                                      stats=st, out=y, tile=args.tile)
        run(); run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        flops = 2.0 * B * Ho * Ho * Cout * k * k * Cin
        byts = (B * H * H * Cin + B * Ho * Ho * Cout) * es
        hbm_us, mf_us = byts / 5.0e12 * 1e6, flops / (2.5e15 if es == 2 else 157.3e12) * 1e6
        print(f"{name:22s} {count:2d} {us:9.1f} {flops / us / 1e6:8.1f} {byts / us / 1e3:8.0f} {hbm_us:12.1f} {mf_us:13.1f}")
        tot_t += us * count
        tot_ideal += max(hbm_us, mf_us) * count
    print(f"trunk convs total {tot_t / 1e3:.2f} ms ; max(hbm@5TB/s, mfma-peak) bound {tot_ideal / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
