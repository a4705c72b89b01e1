#!/usr/bin/env python3
"""Throughput of the GPU HO3D input pipeline (row f4) at the bench batch, with the CPU oracle (= the reference's per-sample work,
one core) beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhentropy_amd import synth, ho3d_dataloader as hd

B = int(os.environ.get("B", 256))
base = [synth.ho3d_sample(20 + i, ((i * 37) % 400 - 200, (i * 53) % 300 - 150)) for i in range(8)]
raw = hd.collate_decoded([base[i % 8] for i in range(B)])
aug = torch.as_tensor(hd.draw_aug(B, np.random.RandomState(3))).cuda()
pipe = hd.HO3DBatchPipeline()
for mode, a in (("evaluation", None), ("training (augmentation)", aug)):
    for _ in range(3):
        pipe(raw, a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        img, t = pipe(raw, a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    out_bytes = B * (3 * 256 * 256 * 4 + 2 * 256 * 256 + 256 * 256 * 4)
    src = float((t["crop_size"] * 2).clamp(max=640).pow(2).sum()) * (3 + 3) + B * 120 * 160 * 3
    print(f"{mode}: B={B} {ms:.3f} ms/batch = {B / ms * 1e3:,.0f} img/s; written {out_bytes / 1e6:.0f} MB + source pixels touched ~{src / 1e6:.0f} MB "
          f"-> {(out_bytes + src) / ms / 1e6:.0f} GB/s")
if os.environ.get("CPU", "1") == "1":
    from oracle import ho3d_ref
    p = synth.ho3d_aug_params(1)
    t0 = time.time()
    for i in range(8):
        ho3d_ref.getitem(base[i], p)
    print(f"CPU oracle (numpy, one core, the reference's per-sample work from the decoded arrays on): {(time.time() - t0) / 8 * 1e3:.1f} ms/sample")
