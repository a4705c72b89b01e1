#!/usr/bin/env python3
"""The Glow branch's train step from the trunk feature on (config C2's R = 16,384 rows, B = 256) - for rocprofv3 --kernel-trace --stats:
   cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/prof_glow -- python3 tools/glow_train_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mhentropy_amd import harness, synth, ops
from mhentropy_amd.network import MHEnt
from mhentropy_amd.train import TrainStep

B, N = int(os.environ.get("B", 256)), int(os.environ.get("N", 64))
special, common = harness.mhent_cfgs(backbone="resnet18", tables=synth.mano_tables(0))
special["q_z_giv_i_model"] = os.environ.get("FLOW", "glow")
model = MHEnt(special, **common)
if special["q_z_giv_i_model"] == "glow":
    model.q_z_giv_i.load_state_dict({k: torch.as_tensor(v) for k, v in synth.glow_state(3).items()}, strict=False)
model.q_z_giv_i.compute_dtype = torch.bfloat16
model = model.cuda().train()
_, yn = synth.batch(5, B, with_image=False)
y = {k: torch.as_tensor(v).cuda() for k, v in yn.items()}
f = torch.as_tensor(np.random.default_rng(6).normal(0, 0.5, (B, 512)).astype(np.float32)).cuda()
ts = TrainStep(model)
for i in range(3):
    ts.forward_backward(None, y, N=N, trunk_out=f); ts.optimizer_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for i in range(K):
    ts.forward_backward(None, y, N=N, trunk_out=f); ts.optimizer_step()
torch.cuda.synchronize()
print(f"{special['q_z_giv_i_model']} head + flow train step (eager, no trunk): {(time.perf_counter() - t0) / K * 1e3:.3f} ms")
