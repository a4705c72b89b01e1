#!/usr/bin/env python3
"""What the memory system gives plain streaming passes on this device (the floor for the HBM-bound layers):
fill (write only), copy (read + write) and a read-only reduction over buffers larger than the 256 MiB Infinity Cache."""
import torch

def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n

for mb in (134, 537, 1074):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device="cuda", dtype=torch.float32)[: n // 2].bfloat16() if False else torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    tf, tc, tr = t(lambda: y.fill_(1.0)), t(lambda: y.copy_(x)), t(lambda: x.view(torch.int16).max())
    print(f"{mb:5d} MB  fill {mb / tf / 1e6:6.2f} TB/s   copy {2 * mb / tc / 1e6:6.2f} TB/s (r+w)   read-reduce {mb / tr / 1e6:6.2f} TB/s")
