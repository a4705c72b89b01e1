#!/usr/bin/env python3
"""Micro-benchmark of the fused coupling-stack kernels (f32 and bf16) at the bench shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhentropy_amd import ops, synth

def main():
    B, K, h, steps = int(os.environ.get("B", 256)), int(os.environ.get("K", 64)), 512, 6
    sd = synth.flow_state(1, 45, 512, (h, h), steps)
    ncoup = 2 * steps
    for mode in ("bf16", "f32"):
        packs, b2, wc, bc = [], [], [], []
        for i in range(ncoup):
            for net in ("s", "t"):
                p = f"{net}.{i}."
                f = ops.flow_pack_net_bf16 if mode == "bf16" else ops.flow_pack_net
                packs.append(f(sd[p + "l.0.weight"], sd[p + "l.1.weight"], sd[p + "l.2.weight"]))
                b2.append(sd[p + "l.2.bias"])
        ws = np.concatenate(packs)
        ws = torch.from_numpy(ws.view(np.int16) if mode == "bf16" else ws).cuda()
        b2 = np.stack(b2)
        if mode == "bf16":
            b2 = np.pad(b2, ((0, 0), (0, 19)))
        b2 = torch.from_numpy(b2).cuda()
        mask = torch.from_numpy(sd["mask"]).cuda()
        R = B * K
        z0 = torch.randn(R, 45, device="cuda")
        cond = torch.randn(B, 2 * ncoup, 2, h, device="cuda") * 0.3
        for _ in range(2):
            ops.flow_couplings(z0, cond, ws, b2, mask, B, h, ops.FLOW_FORWARD)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.flow_couplings(z0, cond, ws, b2, mask, B, h, ops.FLOW_FORWARD)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        flops = R * 2 * 2 * ncoup * (48 * h + h * h + h * 48)
        print(f"{mode}: R={R} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  weight stream {ws.numel() * ws.element_size() / 1e6:.1f} MB")

main()
