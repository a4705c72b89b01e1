#!/usr/bin/env python3
"""Bisect the bf16 coupling kernel against the bf16-rounding oracle: zero parts of the nets and report the error pattern."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhentropy_amd import ops, synth
from oracle import flows_ref

def run(tag, h, steps, B, N, zero=(), seed=9):
    sd = synth.flow_state(seed, 45, 512, (h, h), steps)
    ncoup = 2 * steps
    for k in list(sd):
        for z in zero:
            if k.endswith(z):
                sd[k] = np.zeros_like(sd[k])
    packs, b2, wc, bc = [], [], [], []
    for i in range(ncoup):
        for net in ("s", "t"):
            p = f"{net}.{i}."
            packs.append(ops.flow_pack_net_bf16(sd[p + "l.0.weight"], sd[p + "l.1.weight"], sd[p + "l.2.weight"]))
            b2.append(sd[p + "l.2.bias"])
            for j in range(2):
                wc.append(sd[p + f"c.{j}.weight"]); bc.append(sd[p + f"c.{j}.bias"] + sd[p + f"l.{j}.bias"])
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    wstream = dev(np.concatenate(packs).view(np.int16))
    rng = np.random.default_rng(2)
    feat = rng.normal(0, 1, (B, 512)).astype(np.float32)
    z0 = rng.normal(0, 1, (N * B, 45)).astype(np.float32)
    cond = ops.linear(dev(feat), dev(np.concatenate(wc)), dev(np.concatenate(bc))).view(B, 2 * ncoup, 2, h)
    b2d = dev(np.pad(np.stack(b2), ((0, 0), (0, 64 - 45))))
    x, sum_s, logq = ops.flow_couplings(dev(z0), cond, wstream, b2d, dev(sd["mask"]), B, h, ops.FLOW_FORWARD)
    sdt = {k: torch.as_tensor(v) for k, v in sd.items()}
    with torch.no_grad():
        xr, tot = flows_ref.forward_p_logdet_bf16(sdt, torch.as_tensor(z0), torch.as_tensor(feat).repeat(N, 1))
    e = (x.cpu() - xr).abs()
    bad_rows = (e.max(1)[0] > 1e-2 * xr.abs().max()).nonzero().flatten()
    bad_dims = (e.max(0)[0] > 1e-2 * xr.abs().max()).nonzero().flatten()
    print(f"{tag:28s} h={h} steps={steps} B={B} N={N}: max err {e.max():.3e} (scale {xr.abs().max():.2f}); sum_s err {(sum_s.cpu() - tot).abs().max():.3e}; "
          f"bad rows {bad_rows.numel()}/{N * B} {bad_rows[:8].tolist()}; bad dims {bad_dims.tolist()[:48]}")

if __name__ == "__main__":
    B, N = int(os.environ.get("B", 2)), int(os.environ.get("N", 32))
    Z2 = ("l.2.weight",)
    run("W2=0 (bias only)", 512, 1, B, N, Z2)
    run("W2=0,b2=0 (identity)", 512, 1, B, N, Z2 + ("l.2.bias",))
    run("W1=0 (cond1 -> W2)", 512, 1, B, N, ("l.1.weight",))
    run("W0=0 (cond0 -> W1 -> W2)", 512, 1, B, N, ("l.0.weight",))
    run("W0=0,c0=0", 512, 1, B, N, ("l.0.weight", "c.0.weight"))
    run("full, 1 step", 512, 1, B, N)
    run("full, 6 steps", 512, 6, B, N)
    run("full, 6 steps nonuni", 512, 6, 3, 10)
