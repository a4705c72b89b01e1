#!/usr/bin/env python3
"""First-step losses of TrainStep with the one-launch reverse chain of the flow on and off (same model, same batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import harness, synth
from mhentropy_amd.train import TrainStep
B, K = int(os.environ.get("B", 256)), 64
x, yn = synth.batch(0, B, image_size=256)
x = torch.as_tensor(x).cuda(); y = {k: torch.as_tensor(v).cuda() for k, v in yn.items()}
noise = torch.as_tensor(synth.noise(0, K * B)).cuda()
for fused in ("0", "1", "0", "1"):
    os.environ["MHE_FLOW_REV_FUSED"] = fused
    torch.manual_seed(0)
    model = harness.build_mhent(backbone="resnet50", h_dims=(512, 512), num_steps=6, tables=synth.mano_tables(0), compute_dtype=torch.bfloat16).cuda().train()
    ts = TrainStep(model)
    for i in range(3):
        out = ts.step(x, y, noise=noise, N=K); torch.cuda.synchronize()
        print("fused", fused, "step", i, {k: round(float(v), 3) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}, flush=True)
    del ts, model
