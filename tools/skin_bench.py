#!/usr/bin/env python3
"""Time of the full-mesh pass (mhe_mano_verts_f32: pose workspace + skinning) at the metrics pass' row count (B x N = 256 x 200).
MHE_MANO_SKIN_MFMA=0 | 1 selects the kernel (read once per process); MHE_SKIN_DBG: debugging bit mask of mano_skin.hip (development only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mhentropy_amd import ops, synth, mano_pack
R = int(os.environ.get("R", 51200))
t = synth.mano_tables(0)
blob = torch.as_tensor(mano_pack.pack_tables(t["shapedirs"], t["posedirs"], t["v_template"], t["J_regressor"], t["weights"],
                                             t["hands_components"][:45], t["hands_mean"])).cuda()
rng = np.random.default_rng(0)
z = np.zeros((R, 61), np.float32)
z[:, :48] = rng.normal(0, 0.6, (R, 48)); z[:, 48:58] = rng.normal(0, 1, (R, 10)); z[:, 58:] = rng.normal(0, 0.1, (R, 3))
z = torch.as_tensor(z).cuda()
for _ in range(3):
    v = ops.mano_verts(z, blob)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    v = ops.mano_verts(z, blob)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
print(f"MFMA={os.environ.get('MHE_MANO_SKIN_MFMA', '1')} DBG={os.environ.get('MHE_SKIN_DBG', '0')} R={R}: pose + skin {us:7.1f} us, {v.numel() * 4 / us / 1e3:6.1f} GB/s of vertices, checksum {float(v.double().sum()):.6e}")
