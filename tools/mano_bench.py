#!/usr/bin/env python3
"""Time of the loss-pass MANO kernel (mhe_mano_joints_f32) at the bench's row count.  MHE_MANO_FOUR=0 | 1 selects the one-hypothesis-per-wave
kernel or the four-per-wave one (read once per process: run twice for an A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mhentropy_amd import ops, synth, mano_pack
B, K = int(os.environ.get("B", 256)), int(os.environ.get("K", 64))
t = synth.mano_tables(0)
blob = torch.as_tensor(mano_pack.pack_tables(t["shapedirs"], t["posedirs"], t["v_template"], t["J_regressor"], t["weights"],
                                             t["hands_components"][:45], t["hands_mean"])).cuda()
rng = np.random.default_rng(0)
th45 = torch.as_tensor(rng.normal(0, 0.5, (B * K, 45)).astype(np.float32)).cuda()
det = torch.as_tensor(rng.normal(0, 0.3, (B, 16)).astype(np.float32)).cuda()
cu = torch.as_tensor(rng.uniform(-1, 1, (B, 42)).astype(np.float32)).cuda()
vis = torch.as_tensor((rng.random((B, 21)) < 0.7).astype(np.float32)).cuda()
for want in (("log_p", "norms"), ("z", "xyz", "uv", "terms", "log_p", "norms")):
    for _ in range(3):
        o = ops.mano_joints(th45, det, blob, cu, vis, want=want)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        o = ops.mano_joints(th45, det, blob, cu, vis, want=want)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = th45.numel() * 4 + sum(v.numel() * 4 for v in o.values() if v is not None)
    print(f"MHE_MANO_FOUR={os.environ.get('MHE_MANO_FOUR', '1')} R={B * K} want={'+'.join(want):32s} {us:7.1f} us  {nbytes / us / 1e3:6.1f} GB/s  checksum {float(o['log_p'].double().sum()):.6e}")
