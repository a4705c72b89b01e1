"""Every timed launch of ONE eager train step (or, --forward, forward + loss step) at the bench workload (C2), in issue order, with its algorithmic work: where the
weight-gradient and forward-convolution lines of the step stand against the two rooflines.  python tools/train_lines.py [--min-us 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mhentropy_amd import ops, synth
from mhentropy_amd.train import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--min-us", type=float, default=0.0)
ap.add_argument("--only", default="")
ap.add_argument("--forward", action="store_true", help="the forward + loss step (MHEnt.get_loss) instead of the train step")
a = ap.parse_args()
cfg = bench.WORKLOADS["c2"]
B, K = cfg["B"], cfg["K"]
dev = torch.device("cuda", 0)
model, _ = bench.build_model(cfg, "bf16", 0)
model = model.to(dev).train()
x, yn = synth.batch(0, B, image_size=256)
x = torch.as_tensor(x).to(dev)
y = {k: torch.as_tensor(v).to(dev) for k, v in yn.items()}
ops.rng_state(dev, seed=0)
if a.forward:
    step = lambda: model.get_loss(x, y, mods=["uv"], N=K)
else:
    ts = TrainStep(model)
    step = lambda: ts.step(x, y, N=K)
for _ in range(2):
    step()
torch.cuda.synchronize()
ops.KERNEL_TIMES.clear()
ops.TIMING = ops.TIMING_DG = True
step()
torch.cuda.synchronize()
ops.TIMING = False
tot = {}
for name, flops, e0, e1, nbytes in ops.KERNEL_TIMES:
    us = e0.elapsed_time(e1) * 1e3
    t = tot.setdefault(name, [0, 0.0])
    t[0] += 1; t[1] += us
    if us < a.min_us or (a.only and a.only not in name):
        continue
    print(f"{us:8.1f} us  {flops / us / 1e6:7.0f} TF  {nbytes / us / 1e6:6.2f} TB/s  {flops / 1e9:7.1f} GF {nbytes / 1e6:7.1f} MB  "
          f"floor {max(flops / 2.5e15, nbytes / 8e12) * 1e6:6.1f} us  {name.replace('mhe::', '')[:70]}")
print("---- by kernel")
for name, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{us:9.1f} us {n:4d} x  {name}")
