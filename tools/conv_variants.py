#!/usr/bin/env python3
"""A/B timing of conv kernel variants (mhe_conv_desc.tile) on chosen ResNet-50 layer shapes at the bench batch: all variants
of a shape are timed interleaved in one process (rounds x variants), median and min reported.  Ablation builds of the
phase-pipelined kernel (-DMHE_P8_ABLATIONS) are addressed as tile = 8 + 16 * ABL."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops, resnet

SHAPES = {  # name: (H, Cin, Cout, k, stride)
    "l1c3": (64, 64, 256, 1, 1), "l2c3": (32, 128, 512, 1, 1), "l2ds": (64, 256, 512, 1, 2), "l3c2": (16, 256, 256, 3, 1),
    "l3c3": (16, 256, 1024, 1, 1), "l3ds": (32, 512, 1024, 1, 2), "l4c3": (8, 512, 2048, 1, 1), "l4ds": (16, 1024, 2048, 1, 2),
    "l1c2": (64, 64, 64, 3, 1), "l30c1": (32, 512, 256, 1, 1), "l40c1": (16, 1024, 512, 1, 1), "l20c1": (64, 256, 128, 1, 1), "l1c1": (64, 256, 64, 1, 1), "l2c1": (32, 512, 128, 1, 1), "l4c2": (8, 512, 512, 3, 1), "flowhh": (8, 512, 512, 1, 1), "l3c1": (16, 1024, 256, 1, 1), "l4c1": (8, 2048, 512, 1, 1), "l2c2": (32, 128, 128, 3, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="l3c2,l4ds,l3c3,l1c3")
    ap.add_argument("--tiles", default="0,3,8")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--tail", action="store_true", help="residual-block tail on the operand load (MODE 2): x = relu(bn(x) + x2), also written out")
    ap.add_argument("--bn-load", action="store_true", help="producer BatchNorm + ReLU on the operand load (MODE 1; register-staged tiles only)")
    args = ap.parse_args()
    tiles = [int(t) for t in args.tiles.split(",")]
    B = args.batch
    for name in args.shapes.split(","):
        H, Cin, Cout, k, stride = SHAPES[name]
        x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
        w = resnet.pack_conv_weight(torch.randn(Cout, Cin, k, k) * 0.05, torch.bfloat16).cuda()
        pad = k // 2
        Ho = (H + 2 * pad - k) // stride + 1
        y = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=torch.bfloat16)
        st = None if args.no_stats else ops.stat_unit(Cout, "cuda")
        flops = 2.0 * B * Ho * Ho * Cout * k * k * Cin
        nbytes = 2.0 * (x.numel() + y.numel() + w.numel())
        if args.tail:
            x2, aout = torch.randn_like(x), torch.empty_like(x)
            tsc, tsh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
            nbytes = 2.0 * (3 * x.numel() + y.numel() + w.numel())
        bn = dict(in_scale=torch.rand(Cin, device="cuda") + 0.5, in_shift=torch.randn(Cin, device="cuda") * 0.1, relu_in=True) if args.bn_load else {}
        times = {t: [] for t in tiles}
        for r in range(args.rounds + 1):
            for t in tiles:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    if args.tail:
                        ops.conv1x1_residual_in(x, x2, w, tsc, tsh, a_out=aout, stats=st, tile=t)
                    else:
                        ops.conv2d_nhwc(x, w, k, k, stride, pad, stats=st, out=y, tile=t, **bn)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[t].append(e0.elapsed_time(e1) * 1e3 / args.iters)
        print(f"{name}: H={H} {Cin}->{Cout} k{k} s{stride}  M={B * Ho * Ho}  {flops / 1e9:.1f} GFLOP")
        for t in tiles:
            med, mn = statistics.median(times[t]), min(times[t])
            print(f"   tile {t:4d}: median {med:8.1f} us  min {mn:8.1f} us   {flops / med / 1e6:7.1f} TF/s  {nbytes / med / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
