#!/usr/bin/env python3
"""Body-model path (SURVEY.md section 8 row f1 / config C4 per GPU: B=128 images x K=128 hypotheses of a 144-D 6D pose): time of the
Glow sampling pass, the 6D -> R conversion and the SMPL-sized linear-blend skinning (24 joints, 6,890 vertices), whole and for a
1/8 hypothesis slice (what one rank of a hypothesis-sharded 8-GPU job decodes).  Synthetic tables; parity unpinned at this size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mhentropy_amd import body, synth

B, K = int(os.environ.get("B", 128)), int(os.environ.get("K", 128))
head = body.BodyFlowHead(body.synthetic_body_tables(0), context_features=2048, hidden=1024, num_layers=4, num_blocks=2)
head.flow.load_state_dict({k: torch.as_tensor(v) for k, v in synth.glow_state(1, 144, 1024, 4, 2, 2048).items()}, strict=False)
head = head.cuda().eval()
head.flow.compute_dtype = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
feats = torch.randn(B, 2048, device="cuda") * 0.5
betas = torch.randn(B, 10, device="cuda")
noise = torch.randn(B, K, 144, device="cuda")


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    pose, logp, _ = head.flow.sample_and_log_prob(K, noise=noise, context=feats)
    p = pose.reshape(B * K, 144).contiguous()
    bt = betas[:, None, :].expand(B, K, 10).reshape(B * K, 10).contiguous()
    R = B * K
    ms_flow = t(lambda: head.flow.sample_and_log_prob(K, noise=noise, context=feats))
    ms_rot = t(lambda: body.rot6d_to_rotmat(p.view(R, 24, 6)))
    rm = body.rot6d_to_rotmat(p.view(R, 24, 6))
    ms_lbs = t(lambda: head.body(bt, rotmats=rm))
    ms_joints = t(lambda: head.body(bt, rotmats=rm, want_verts=False))
    ms_all = t(lambda: head(feats, K, betas=betas, noise=noise))
    ms_slice = t(lambda: head(feats, K, betas=betas, noise=noise, hyp_slice=(0, K // 8)))
vb = R * 6890 * 12
print(f"B={B} K={K} R={R}: glow sample+log_prob {ms_flow:.2f} ms | rot6d {ms_rot * 1e3:.0f} us | LBS 6,890 verts {ms_lbs:.2f} ms "
      f"({vb / ms_lbs / 1e6:.0f} GB/s of vertices written, {R * 6890 * 3 * (10 + 207 + 96 + 4) / ms_lbs / 1e9:.1f} TFMA/s) | joints only {ms_joints * 1e3:.0f} us")
print(f"whole head {ms_all:.2f} ms = {R / ms_all * 1e3:.3e} hypotheses/s ; decoding a 1/8 hypothesis slice {ms_slice:.2f} ms")
