#!/usr/bin/env python3
"""time mhe_bottleneck_tail256_nhwc at config C2's layer3 shape (B = 256, 16 x 16, bottleneck 256) from a stand-alone library
(MHE_T256_LIB: an ablation build of tools/tail256_abl.sh; default: the product library)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops, _lib

B, H, W = 256, 16, 16
dev = "cuda"
y2 = torch.randn(B, H, W, 256, device=dev).bfloat16()
idt = torch.randn(B, H, W, 1024, device=dev).bfloat16()
w3 = (torch.randn(1024, 256, device=dev) * 0.08).bfloat16()
w1 = (torch.randn(256, 1024, device=dev) * 0.04).bfloat16()
w3s, w1s = ops.bottleneck_tail256_pack(w3, w1)
s2, h2 = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
s3, h3 = torch.rand(1024, device=dev) + 0.5, torch.randn(1024, device=dev) * 0.1
a, y1 = torch.empty_like(idt), torch.empty_like(y2)
st = ops.stat_unit(256, dev)
path = os.environ.get("MHE_T256_LIB")
L = C.CDLL(os.path.abspath(path)) if path else _lib.lib()
fn = L.mhe_bottleneck_tail256_nhwc
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 15
d = ops.ConvDesc(B, H, W, 1024, 256, 1, 1, 1, 0, ops.BF16, 1, 0, 0, 0)
p = lambda t: C.c_void_p(t.data_ptr())
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run():
    rc = fn(C.cast(C.byref(d), C.c_void_p), p(y2), p(s2), p(h2), p(w3s), p(s3), p(h3), p(idt), None, None, p(w1s), p(a), p(y1), p(st), stream)
    assert rc == 0, rc


for _ in range(3):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 100)
ts.sort()
print(f"tail256: median {ts[3]:.1f} us  min {ts[0]:.1f} us   ({335.5e6 / ts[3] / 1e6:.2f} TB/s of compulsory bytes, {68.7e9 / ts[3] / 1e6:.0f} TF/s)")
