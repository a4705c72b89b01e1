#!/usr/bin/env python3
"""Time of the fused reverse chain of the flow (mhe_flow_reverse_chain_bf16) alone at the bench size, random operands.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops
B, N, h, ncoup, dim = int(os.environ.get("B", 256)), 64, 512, 12, 45
R, nets = N * B, 2 * ncoup
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
x_out, g_x, g_logp = rn(R, dim), rn(R, dim) * 0.01, rn(B) * 0.01
mask = torch.zeros(ncoup, dim, device="cuda"); mask[0::2, :22] = 1; mask[1::2, 22:] = 1
o_pre = rn(nets, R, 64) * 0.1
h1, h2 = rn(nets, R, h).bfloat16(), rn(nets, R, h).bfloat16()
wst = h * h + 2 * 64 * h + 4096
wbuf = (torch.randn(nets * wst + h * h, device="cuda", generator=g) * 0.03).bfloat16()
w1T, w2T, w0T = wbuf, wbuf[h * h:], wbuf[h * h + 64 * h:]          # (fragment-major or not: the timing does not care)
GOb, XPb = torch.empty(nets, R, 64, device="cuda", dtype=torch.bfloat16), torch.empty(ncoup, R, 64, device="cuda", dtype=torch.bfloat16)
G2b, G1b = torch.empty(nets, R, h, device="cuda", dtype=torch.bfloat16), torch.empty(nets, R, h, device="cuda", dtype=torch.bfloat16)
cs = 4 * ncoup * h
Gc, db2, z0 = torch.zeros(B, cs, device="cuda"), torch.zeros(nets * 64, device="cuda"), torch.empty(R, dim, device="cuda")
sg = ops.flow_sign_bits(h1, h2, B)
run = lambda: ops.flow_reverse_chain(x_out, g_x, g_logp, -1.0 / N, mask, o_pre, sg, w2T, w1T, w0T, wst, GOb, G2b, G1b, XPb, Gc, db2, 64, z0)
for _ in range(2): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print(f"chain kernel {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us  (R = {R}, {nets} nets)")
