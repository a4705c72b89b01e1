#!/usr/bin/env python3
"""Timing of the reference's per-step metrics pass (hand/CrossModalHand.py:357-361, criteria.py):
MHEnt.sample(N=[N,N], temp=0.8, mods={'uv','xyz','verts'}) + MHEntLoss, from the conditioning feature on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import harness, synth, ops
from mhentropy_amd.criteria import MHEntLoss
B, N = int(os.environ.get("B", 256)), int(os.environ.get("N", 200))
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
model = harness.build_mhent(backbone="resnet50", tables=synth.mano_tables(0), compute_dtype=dt).cuda().eval()
_, yn = synth.batch(0, B, with_image=False)
y = {k: torch.as_tensor(v).cuda() for k, v in yn.items()}
feat = torch.randn(B, 512, device="cuda") * 0.5
model.feat_extractor.forward = lambda x: (feat, feat, None)          # time the decoder side only
crit = MHEntLoss()
lp = torch.zeros(B, device="cuda")
def run():
    s = model.sample(None, N=[N, N], temp=0.8, mods={"uv", "xyz", "verts"}, y=y)
    s["log_p"] = lp
    return crit(s, y)
for _ in range(2): run()
torch.cuda.synchronize()
ops.KERNEL_TIMES.clear()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): run()
e1.record(); torch.cuda.synchronize()
print(f"sample(N={N}) + metrics, B={B}: {e0.elapsed_time(e1) / 3:.2f} ms  ({B * N / (e0.elapsed_time(e1) / 3e3):.3e} hypotheses/s), verts bytes {B * N * 2334 * 4 / 1e6:.0f} MB")
