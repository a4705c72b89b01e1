#!/usr/bin/env python3
"""Reverse pass of the trunk twice over one forward tape: conv3 + bn3 of layer1 / layer2 reversed by reading y3 (B) and on the Gram statistics (C),
run-to-run (B2, C2): per-tensor norm-wise differences of the gradients, last block first."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from mhentropy_amd import harness, synth
from mhentropy_amd.train import TrainStep
torch.manual_seed(3)
S = int(os.environ.get("S", 128)); B = int(os.environ.get("B", 8))
model = harness.build_mhent(backbone="resnet50", h_dims=(64, 64), num_steps=1, tables=synth.mano_tables(0), compute_dtype=torch.bfloat16).cuda().train()
N = 4
xn, yn = synth.batch(9, B, image_size=S)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
x, y = dev(xn), {k: dev(v) for k, v in yn.items()}
z0 = dev(synth.noise(9, N * B))
ts = TrainStep(model)
ts.train_recompute, ts.conv3_fold = True, True
ts.forward(x, y, noise=z0, N=N)
res = {}
for mode, fold in (("B", False), ("B2", False), ("C", True), ("C2", True)):
    ts.conv3_fold = fold
    ts.backward()
    res[mode] = {n: ts.grad_of(p).clone() for n, p in model.named_parameters()}
    print(mode, "folds", ts.n_fold)
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
names = [n for n in res["B"] if n.startswith("feat_extractor.res.layer") and res["B"][n].abs().max() > 0]
order = []
for L in (4, 3, 2, 1):
    nb = {4: 3, 3: 6, 2: 4, 1: 3}[L]
    for bi in range(nb - 1, -1, -1):
        for part in ("conv3.weight", "bn3.weight", "bn3.bias", "conv2.weight", "bn2.weight", "conv1.weight", "bn1.weight"):
            order.append(f"feat_extractor.res.layer{L}.{bi}.{part}")
for n in order:
    if n in res["B"] and (".layer2." in n or ".layer1." in n or n.startswith("feat_extractor.res.layer3.0")):
        print("%-50s B2/B %.2e  C/B %.2e  C2/C %.2e" % (n[19:], rel(res["B2"][n], res["B"][n]), rel(res["C"][n], res["B"][n]), rel(res["C2"][n], res["C"][n])))
