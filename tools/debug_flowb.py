import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mhentropy_amd import ops, synth
from oracle import flows_ref
h, steps, B, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sd = synth.flow_state(9, 45, 512, (h, h), steps)
ncoup = 2 * steps
if os.environ.get('ZERO_B2'):
    for k in sd:
        if k.endswith('l.2.bias'): sd[k] = sd[k] * 0
if os.environ.get('ZERO_B2S'):
    for k in sd:
        if k.endswith('l.2.bias') and k.startswith('s.'): sd[k] = sd[k] * 0
if os.environ.get('ZERO_B2T'):
    for k in sd:
        if k.endswith('l.2.bias') and k.startswith('t.'): sd[k] = sd[k] * 0
if os.environ.get('ZERO_COND'):
    for k in sd:
        if '.c.' in k or k.endswith('l.0.bias') or k.endswith('l.1.bias'): sd[k] = sd[k] * 0
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
b2d = dev(np.pad(np.stack(b2), ((0, 0), (0, 19))))
ncs = int(sys.argv[5]) if len(sys.argv) > 5 else ncoup
mask = sd["mask"][:ncs]
x, sum_s, logq = ops.flow_couplings(dev(z0), cond[:, :2 * ncs].contiguous(), wstream, b2d[:2 * ncs].contiguous(), dev(mask), B, h, ops.FLOW_FORWARD)
sdt = {k: torch.as_tensor(v) for k, v in sd.items()}
sdt["mask"] = torch.as_tensor(mask)
with torch.no_grad():
    xr, tot = flows_ref.forward_p_logdet_bf16(sdt, torch.as_tensor(z0), torch.as_tensor(feat).repeat(N, 1))
e = (x.cpu() - xr).abs()
print("x err dims 0..7:", np.round((x.cpu() - xr)[0, :8].numpy(), 4), "t-bias", np.round(sd["t.0.l.2.bias"][:8], 4), "s-bias", np.round(sd["s.0.l.2.bias"][:8], 4))
print("max err per dim:", np.round(e.max(0)[0].numpy(), 3))
print("max err per row (first 16):", np.round(e.max(1)[0].numpy()[:16], 3))
print("sum_s err", (sum_s.cpu() - tot).abs().max().item())
x2, s2, _ = ops.flow_couplings(dev(z0), cond[:, :2 * ncs].contiguous(), wstream, b2d[:2 * ncs].contiguous(), dev(mask), B, h, ops.FLOW_FORWARD)
print("forward deterministic:", torch.equal(x, x2), (x - x2).abs().max().item())
zb, s3, _ = ops.flow_couplings(x, cond[:, :2 * ncs].contiguous(), wstream, b2d[:2 * ncs].contiguous(), dev(mask), B, h, ops.FLOW_INVERSE)
print("inverse err", (zb.cpu() - torch.as_tensor(z0)).abs().max().item(), "sum_s diff", (s3 - sum_s).abs().max().item())
e2 = (zb.cpu() - torch.as_tensor(z0)).abs()
print("rows with err>1e-4:", (e2.max(1)[0] > 1e-4).nonzero().flatten().tolist()[:40])
