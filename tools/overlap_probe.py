"""Do two independent launches of a stage's first block - the shortcut's 1x1 / stride-2 convolution (HBM-bound) and conv2, 3x3 / stride 2
with the producer's BatchNorm on its load (matrix-core / latency-bound) - overlap when issued on two streams?  Times at config C2's shapes:
back to back on one stream against forked on two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops, resnet

def run(B, H, Cin_ds, Cout_ds, Cb):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, H, H, Cin_ds, device=dev, generator=g).bfloat16()
    y1 = torch.randn(B, H, H, Cb, device=dev, generator=g).bfloat16()
    wd = resnet.pack_conv_weight(torch.randn(Cout_ds, Cin_ds, 1, 1) * 0.05, torch.bfloat16).cuda()
    w2 = resnet.pack_conv_weight(torch.randn(Cb, Cb, 3, 3) * 0.05, torch.bfloat16).cuda()
    sc, sh = torch.rand(Cb, device=dev) + 0.5, torch.randn(Cb, device=dev) * 0.1
    od = torch.empty(B, H // 2, H // 2, Cout_ds, device=dev, dtype=torch.bfloat16)
    o2 = torch.empty(B, H // 2, H // 2, Cb, device=dev, dtype=torch.bfloat16)
    st_d, st_2 = ops.stat_unit(Cout_ds, dev), ops.stat_unit(Cb, dev)
    f_ds = lambda: ops.conv2d_nhwc(x, wd, 1, 1, 2, 0, stats=st_d, out=od)
    f_c2 = lambda: ops.conv2d_nhwc(y1, w2, 3, 3, 2, 1, in_scale=sc, in_shift=sh, relu_in=True, stats=st_2, out=o2)
    side = torch.cuda.Stream()
    def serial():
        f_ds(); f_c2()
    def forked():
        main = torch.cuda.current_stream()
        e = torch.cuda.Event(); e.record(main)
        with torch.cuda.stream(side):
            side.wait_event(e)
            f_ds()
            e2 = torch.cuda.Event(); e2.record(side)
        f_c2()
        main.wait_event(e2)
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    print(f"B={B} H={H} ds {Cin_ds}->{Cout_ds}, conv2 {Cb}: ds alone {timeit(f_ds):.1f} us, conv2 alone {timeit(f_c2):.1f} us, "
          f"serial {timeit(serial):.1f} us, forked {timeit(forked):.1f} us")

run(256, 64, 256, 512, 128)       # layer2.0
run(256, 32, 512, 1024, 256)      # layer3.0
run(256, 16, 1024, 2048, 512)     # layer4.0 (conv2 there runs on a pre-normalised input; here with BatchNorm on load)
