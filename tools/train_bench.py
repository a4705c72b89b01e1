#!/usr/bin/env python3
"""Wall time of TrainStep.step (forward + reverse + clip + Adam) at a bench workload; eager or graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import harness, synth
from mhentropy_amd.train import TrainStep
B, K = int(os.environ.get("B", 256)), int(os.environ.get("K", 64))
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
bb, h, steps = os.environ.get("BB", "resnet50"), int(os.environ.get("H", 512)), int(os.environ.get("STEPS", 6))
model = harness.build_mhent(backbone=bb, h_dims=(h, h), num_steps=steps, tables=synth.mano_tables(0), compute_dtype=dt).cuda().train()
x, yn = synth.batch(0, B, image_size=int(os.environ.get("S", 256)))
x = torch.as_tensor(x).cuda(); y = {k: torch.as_tensor(v).cuda() for k, v in yn.items()}
noise = torch.as_tensor(synth.noise(0, K * B)).cuda()
ts = TrainStep(model)
for i in range(2):
    out = ts.step(x, y, noise=noise, N=K); torch.cuda.synchronize()
    print("warm", i, float(out["total"]), flush=True)
if os.environ.get("TRACE"):
    for i in range(int(os.environ["TRACE"])):
        out = ts.step(x, y, noise=noise, N=K)
        if i % 5 == 0: print("step", i, float(out["total"]), flush=True)
n = int(os.environ.get("ITERS", 3))
run = lambda: ts.step(x, y, noise=noise, N=K)
if os.environ.get("GRAPH") == "1":
    gs = torch.cuda.Stream(); gs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(gs):
        ts.step(x, y, noise=noise, N=K)
    torch.cuda.current_stream().wait_stream(gs)
    graph = torch.cuda.CUDAGraph(); res = {}
    with torch.cuda.graph(graph):
        res["out"] = ts.step(x, y, noise=noise, N=K)
    graph.replay(); torch.cuda.synchronize()
    print("graph captured; loss", float(res["out"]["total"]), flush=True)
    def run():
        graph.replay(); return res["out"]
t0 = time.time()
for _ in range(n):
    out = run()
torch.cuda.synchronize()
dtm = (time.time() - t0) / n
print(f"train step {bb} B={B} K={K} {os.environ.get('DT', 'bf16')}: {dtm * 1e3:.2f} ms/step  {B / dtm:.1f} img/s  loss {float(out['total']):.4f}", flush=True)
