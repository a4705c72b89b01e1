"""csrc/conv_halo.hip against the pass + im2col path it replaces: values and time.  python tools/halo_check.py [B]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mhentropy_amd import ops, resnet

ABL = os.environ.get("MHE_HALO_LIB")        # a tuning build (tools/halo_abl_build.sh): timed instead of the library's kernel
if ABL:
    import ctypes as ct
    _abl = ct.CDLL(ABL)
    _abl.mhe_conv3x3_halo_nhwc.restype = ct.c_int
    _abl.mhe_conv3x3_halo_nhwc.argtypes = [ct.c_int] * 5 + [ct.c_void_p] * 5 + [ct.c_int] + [ct.c_void_p] * 8
    def _halo(x, wh, sc=None, sh=None, relu_in=False, a_out=None, stats=None):
        B, H, W, Cin = x.shape
        y = torch.empty(B, H, W, wh.shape[0], device=x.device, dtype=torch.bfloat16)
        q = lambda t: None if t is None else ct.c_void_p(t.data_ptr())
        rc = _abl.mhe_conv3x3_halo_nhwc(B, H, W, Cin, wh.shape[0], q(x), q(wh), q(y), q(sc), q(sh), int(relu_in), q(a_out), q(stats), None, None, None, None, None,
                                        ct.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        return y
    ops.conv3x3_halo = _halo


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (H, C) in ((32, 128), (16, 256)):
    W = H
    x = (torch.randn(B, H, W, C, device=dev) * 1.5).bfloat16()
    wt = torch.randn(C, C, 3, 3, device=dev) * (1.0 / (9 * C) ** 0.5)
    w = resnet.pack_conv_weight(wt, torch.bfloat16)
    wh = ops.conv3x3_halo_pack(w)
    sc = (torch.rand(C, device=dev) + 0.5).contiguous(); sh = (torch.randn(C, device=dev) * 0.3).contiguous()
    S = ops.stat_shards()
    st0 = ops.stat_unit(C, "cuda"); st1 = ops.stat_unit(C, "cuda")
    # the path it replaces: in-place pass + im2col conv
    a_ref = ops.bn_act(x.clone(), sc, sh, relu=True)
    y_ref = ops.conv2d_nhwc(a_ref, w, 3, 3, 1, 1, stats=st0)
    a_out = torch.empty_like(x)
    y = ops.conv3x3_halo(x, wh, sc, sh, relu_in=True, a_out=a_out, stats=st1)
    torch.cuda.synchronize()
    print(f"H={H} C={C}: a_out == pass: {torch.equal(a_out, a_ref)}; max|y - y_ref| = {(y.float() - y_ref.float()).abs().max().item():.4g} "
          f"(max|y| {y_ref.float().abs().max().item():.3g}); differing {(y != y_ref).float().mean().item():.2e}; "
          f"stats rel {((st1.sum(0) - st0.sum(0)).abs().max() / st0.sum(0).abs().max()).item():.2e}")
    # torch f32 reference on the same rounded operands
    if B <= 32:
        yt = torch.nn.functional.conv2d(a_ref.float().permute(0, 3, 1, 2), wt.bfloat16().float(), padding=1).permute(0, 2, 3, 1)
        print("   vs torch f32:", (y.float() - yt).abs().max().item(), "  old path:", (y_ref.float() - yt).abs().max().item())
    t_pass = timeit(lambda: ops.bn_act(x, sc, sh, relu=True, out=a_out))
    t_conv = timeit(lambda: ops.conv2d_nhwc(a_ref, w, 3, 3, 1, 1, stats=st0))
    t_halo = timeit(lambda: ops.conv3x3_halo(x, wh, sc, sh, relu_in=True, stats=st1))
    t_halo_a = timeit(lambda: ops.conv3x3_halo(x, wh, sc, sh, relu_in=True, a_out=a_out, stats=st1))
    t_plain = timeit(lambda: ops.conv3x3_halo(a_ref, wh, stats=st1))
    fl = 2.0 * B * H * W * C * C * 9
    print(f"   pass {t_pass:.1f} us + conv {t_conv:.1f} us = {t_pass + t_conv:.1f};  halo {t_halo:.1f} us ({fl / t_halo / 1e6:.0f} TFLOP/s), with a_out {t_halo_a:.1f}, plain operand {t_plain:.1f}")
