#!/usr/bin/env python3
"""3x3 / stride-2 data gradient at the three ResNet-50 shapes (C2 batch): parity classes vs the zero-dilated form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops, train

def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

B = int(os.environ.get("B", 256))
for C, H in ((128, 64), (256, 32), (512, 16)):
    dt = torch.bfloat16
    w = torch.randn(C, C, 3, 3) / (C * 9) ** 0.5
    idx = torch.arange(w.numel()).view(w.shape)
    pack = lambda tb: torch.nn.functional.pad(w.reshape(-1)[tb], (0, (-tb.shape[1]) % 64)).to(dt).cuda().contiguous()
    w_s2 = [pack(tb) for tb in train.dgrad_s2_operand_indices(idx)]
    wd = pack(train.dgrad_operand_index(idx))
    gy = torch.randn(B, H // 2, H // 2, C, device="cuda").to(dt)
    mask = torch.randn(B, H, H, C, device="cuda").to(dt)
    res = torch.randn(B, H, H, C, device="cuda").to(dt)
    bn_y = torch.randn(B, H, H, C, device="cuda").to(dt)
    mi = torch.rand(2, C, device="cuda") + 0.5
    st = ops.stat_unit(C, "cuda")
    new = t(lambda: ops.conv3x3s2_dgrad(gy, w_s2, residual=res, mask=mask, bn=[(bn_y, mi, st)]))
    old = t(lambda: ops.conv2d_nhwc(ops.upsample2(gy, H, H), wd, 3, 3, 1, 1, residual=res, mask=mask, bn=[(bn_y, mi, st)]))
    print(f"C={C} H={H}: parity classes {new:8.1f} us   zero-dilated {old:8.1f} us", flush=True)
    for tile in (1, 2, 3, 8):
        try:
            print(f"    tile {tile}: {t(lambda: ops.conv3x3s2_dgrad(gy, w_s2, residual=res, mask=mask, bn=[(bn_y, mi, st)], tile=tile)):8.1f} us")
        except Exception as e:
            print("    tile", tile, "n/a", str(e)[:60])
