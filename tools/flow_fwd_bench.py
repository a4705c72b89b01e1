#!/usr/bin/env python3
"""Time of the coupling stack alone at the bench size: second-generation kernel (mhe_flow_couplings_bf16) and fragment-streaming kernel
(mhe_flow_couplings_frag_bf16), plain and with the activations written out."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhentropy_amd import ops, synth
B, N, h, steps = int(os.environ.get("B", 256)), 64, 512, 6
ncoup, R = 2 * steps, 64 * int(os.environ.get("B", 256))
sd = synth.flow_state(9, 45, 512, (h, h), steps)
per, packs, b2 = [], [], []
for i in range(ncoup):
    for net in ("s", "t"):
        p = f"{net}.{i}."
        w = [sd[p + f"l.{j}.weight"] for j in range(3)]
        f1, f0, f2 = ops.flow_frag_pack(*(torch.as_tensor(x) for x in w))
        per.append(torch.cat([f1.reshape(-1), f0.reshape(-1), f2.reshape(-1)]))
        packs.append(ops.flow_pack_net_bf16(*w)); b2.append(sd[p + "l.2.bias"])
fp = torch.stack(per).to(torch.bfloat16).cuda().contiguous()
w0F, w1F, w2F, pitch = fp[0, h * h:], fp[0], fp[0, h * h + 64 * h:], fp.shape[1]
ws = torch.as_tensor(np.concatenate(packs).view(np.int16)).cuda()
b2d = torch.as_tensor(np.pad(np.stack(b2), ((0, 0), (0, 64 - 45)))).cuda()
mask = torch.as_tensor(sd["mask"]).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
z0 = torch.randn(R, 45, device="cuda", generator=g)
cond = torch.randn(B, 2 * ncoup, 2, h, device="cuda", generator=g) * 0.5
keep = (torch.zeros(2 * ncoup, R, h, device="cuda", dtype=torch.bfloat16), torch.zeros(2 * ncoup, R, h, device="cuda", dtype=torch.bfloat16),
        torch.zeros(2 * ncoup, R, 64, device="cuda"))
def timeit(name, fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us")
timeit("second generation (LDS-DMA rings)", lambda: ops.flow_couplings(z0, cond, ws, b2d, mask, B, h, ops.FLOW_FORWARD))
timeit("fragment streaming", lambda: ops.flow_couplings_frag(z0, cond, w0F, w1F, w2F, pitch, b2d, mask, B, h, ops.FLOW_FORWARD))
timeit("second generation, activations written out", lambda: ops.flow_couplings_emit(z0, cond, ws, b2d, mask, B, h, ops.FLOW_FORWARD, *keep))
timeit("fragment streaming, activations written out", lambda: ops.flow_couplings_frag(z0, cond, w0F, w1F, w2F, pitch, b2d, mask, B, h, ops.FLOW_FORWARD, emit=keep))
