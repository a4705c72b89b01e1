#!/usr/bin/env python3
"""Per-parameter gradient error of TrainStep vs autograd on the oracle (diagnostic; GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from mhentropy_amd import synth
from mhentropy_amd.train import TrainStep
from oracle import train_ref, mano_ref
from test_gpu_train import _model_and_state
bb, h, steps = os.environ.get("BB", "resnet50"), int(os.environ.get("H", 512)), int(os.environ.get("STEPS", 6))
B, N, S = int(os.environ.get("B", 4)), int(os.environ.get("N", 4)), int(os.environ.get("S", 128))
model, sd = _model_and_state(bb, h, steps)
tb = mano_ref.tables_from_numpy(synth.mano_tables(0))
xn, yn = synth.batch(7, B, image_size=S)
z0 = torch.as_tensor(synth.noise(7, N * B))
x, y = torch.as_tensor(xn), {k: torch.as_tensor(v) for k, v in yn.items()}
DT = torch.float64 if os.environ.get("F64", "1") == "1" else torch.float32
tb = mano_ref.tables_from_numpy(synth.mano_tables(0), dtype=DT)
cast = lambda d: {k: (v.to(DT) if v.is_floating_point() else v) for k, v in d.items()}
out_ref, total_ref, grads, _ = train_ref.loss_and_grads(cast(sd), tb, x.to(DT), cast(y), z0.to(DT), N, arch=bb)
grads = {k: v.float() for k, v in grads.items()}
ts = TrainStep(model)
out = ts.forward_backward(x.cuda(), {k: v.cuda() for k, v in y.items()}, noise=z0.cuda(), N=N)
print("log_p", out["log_p"].cpu(), out_ref["log_p"].detach())
rows = []
for name, p in model.named_parameters():
    if name in grads:
        got, want = ts.grad_of(p).cpu(), grads[name]
        sc = want.abs().max().item()
        rows.append(((got - want).abs().max().item() / (sc + 1e-30), name, sc))
for r in rows:
    print(f"{r[0]:.2e}  {r[2]:.3e}  {r[1]}")
