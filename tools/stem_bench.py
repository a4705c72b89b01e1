import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops, resnet
for dt in (torch.bfloat16, torch.float32):
    B = 256
    x = torch.randn(B, 3, 256, 256, device="cuda")
    w = resnet.pack_stem_weight(torch.randn(64, 3, 7, 7) * 0.1, dt).cuda()
    st = ops.stat_unit(64, "cuda")
    for _ in range(2): ops.stem_conv7x7s2(x, w, dt, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): y = ops.stem_conv7x7s2(x, w, dt, stats=st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    byts = x.numel() * 4 + y.numel() * y.element_size()
    print(dt, f"{us:8.1f} us  {byts / us / 1e3:7.0f} GB/s (algorithmic)  ideal@5TB/s {byts / 5e12 * 1e6:.0f} us")
