"""Tensor-level wrappers over the C ABI (include/mhe.h).

torch is plumbing here: it owns device memory and the HIP stream; every number is
produced by the hand-written kernels in csrc/.  All functions require CUDA (HIP)
tensors and raise on anything else - there is no CPU path in the product.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, check

F32, BF16 = 0, 1
FLOW_FORWARD, FLOW_INVERSE = 0, 1

# bench.py instrumentation: when TIMING is set every conv launch is bracketed by
# HIP events on the launch stream and logged as (kernel name, algorithmic flops, ev0, ev1)
TIMING = False
TIMING_DG = False        # tools/train_lines.py: also the data-gradient launches, under descriptive names (not rocprofv3's: bench.py leaves it off)
KERNEL_TIMES = []


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk(t, dtype, name, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.MheError(f"{name}: expected a CUDA/HIP tensor (the hot path has no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.MheError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.MheError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.MheError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def stat_shards():
    return _lib.lib().mhe_conv_stat_shards()


STAT_DTYPE = torch.int64


def stat_shape(C):
    """one unit of sharded fixed-point accumulators for C channels (include/mhe.h, mhe_stat_t): [2 planes][S shards][2 statistics][C]"""
    return (2, stat_shards(), 2, C)


def stat_unit(C, device):
    return torch.zeros(stat_shape(C), device=device, dtype=STAT_DTYPE)


def stat_totals(st):
    """[2, C] float64 totals of a unit: sum over shards of plane0 * 2^-16 + plane1 * 2^-56 (what the finalize kernels compute; the
    marker of a non-finite partial is not decoded here - tests and diagnostics only)"""
    t = st.sum(1)                                     # integer sums: exact
    return t[0].double() * 2.0 ** -16 + t[1].double() * 2.0 ** -56


def stat_from_float(t):
    """[S, 2, C] float shard values -> a unit holding them (the encoding fx::add2 of csrc/common.h applies to one partial; tests)"""
    d = t.double() * 2.0 ** 16
    hi = torch.round(d)
    lo = torch.round((d - hi) * 2.0 ** 40)
    return torch.stack([hi.to(torch.int64), lo.to(torch.int64)]).contiguous()


def _chk_stats(t, name, C):
    _chk(t, STAT_DTYPE, name, stat_shape(C))


_TILES = {0: "128, 64, 2, 2", 1: "128, 128, 2, 2", 2: "256, 256, 2, 4", 3: "256, 128, 4, 2", 4: "256, 64, 4, 2"}


def _conv_kernel_name(d, dt, mode):
    """the template instantiation the launcher will pick, spelled as rocprofv3 prints it"""
    bke = 32 if dt == torch.float32 else 64
    tile = _lib.lib().mhe_conv_tile_mode(C.byref(d), mode)
    mode = 0 if mode == 3 else mode          # (3 = plain operand load + a residual in the epilogue: the kernel's MODE is 0)
    if tile == 11:
        return "mhe::conv::conv_wide_kernel<%s, %s>" % ("128, 4" if d.Cin == 256 else "64, 8", "true" if mode == 1 else "false")
    if tile == 10:
        return "mhe::conv::conv_tail_kernel<false>"
    if tile == 9:
        return "mhe::conv::conv3x3_c64_stream_kernel<%s, false>" % ("true" if mode == 1 else "false")
    if tile == 8:
        return "mhe::conv::conv1x1_stream_kernel<%d, %d, %s, false>" % (d.Cin // 64, 64 if (d.Cin == 64 and d.Cout == 64) else 256 if d.Cin <= 128 else 128, "true" if mode == 1 else "false")
    if tile == 13:
        return "mhe::conv::conv_p8_kernel<false, %s, 0, 1>" % ("false" if d.KH == 1 and d.KW == 1 and d.pad == 0 else "true")
    if tile == 7:
        return "mhe::conv::conv_p8_kernel<false, %s, 0, 2>" % ("false" if d.KH == 1 and d.KW == 1 and d.pad == 0 else "true")
    return "mhe::conv::conv_kernel<%s, %s, %s, %d, false>" % ("float" if dt == torch.float32 else "unsigned short",
                                                             _TILES[tile],
                                                             "true" if d.Cin % bke == 0 else "false", mode)


def dtype_code(dt):
    return F32 if dt == torch.float32 else BF16


def conv_tile_choice(B, H, W, Cin, Cout, k, stride, pad, dt, mode=0):
    """the tile variant the launcher picks for this convolution (mhe_conv_tile_mode; include/mhe.h lists the variants).  mode 1 = with the
    producer's BatchNorm on the operand load.  9 = the row-streaming 3x3 kernel, whose on-load form costs one multiply-add per element read
    ONCE (rows are normalised on their way into the LDS ring), unlike the tiled kernels that normalise every tap's re-read."""
    d = ConvDesc(B, H, W, Cin, Cout, k, k, stride, pad, dtype_code(dt), int(mode == 1), 0, 0, 0)
    return _lib.lib().mhe_conv_tile_mode(C.byref(d), mode)


def linear(x, w, bias=None, relu=False, out=None, want_bf16=False):
    """act(x @ w.T + bias) in f32 (mhe_linear_f32).  want_bf16: returns (y, y as bf16) from the same launch where the shape admits it
    (mhe_linear_f32_bf16copy: M <= 256, K % 64 == 0), else (y, None)."""
    M, K = x.shape
    N = w.shape[0]
    _chk(x, torch.float32, "linear.x"); _chk(w, torch.float32, "linear.w", (N, K))
    if bias is not None:
        _chk(bias, torch.float32, "linear.bias", (N,))
    y = out if out is not None else torch.empty(M, N, device=x.device, dtype=torch.float32)
    _chk(y, torch.float32, "linear.out", (M, N))
    if want_bf16:
        if M <= 256 and K % 64 == 0 and N % 4 == 0:
            yb = torch.empty(M, N, device=x.device, dtype=torch.bfloat16)
            check(_lib.lib().mhe_linear_f32_bf16copy(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(yb), M, N, K, int(relu), _stream()), "mhe_linear_f32_bf16copy")
            return y, yb
        check(_lib.lib().mhe_linear_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), M, N, K, int(relu), _stream()), "mhe_linear_f32")
        return y, None
    check(_lib.lib().mhe_linear_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), M, N, K, int(relu), _stream()), "mhe_linear_f32")
    return y


_RNG_STATE = {}


def rng_state(device, seed=None):
    """the device-resident generator state {seed, next counter, 0} of mhe_randn_f32 for `device`; created on first use from torch's
    seed (torch.manual_seed(s) before the first draw makes runs repeatable), or re-seeded explicitly with seed="""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    st = _RNG_STATE.get(device)
    if st is None:
        s = torch.initial_seed() if seed is None else int(seed)
        st = _RNG_STATE[device] = torch.tensor([s & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64, device=device)
    elif seed is not None:
        # re-seed IN PLACE: a captured HIP graph has this tensor's address baked into its randn_kernel node
        st.copy_(torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64))
    return st


def rng_get_state(device):
    """the three int64 words {seed, next counter, draws} of the device generator, on the host (for checkpoints)"""
    return rng_state(device).cpu().clone()


def rng_set_state(device, words):
    """restore rng_get_state()'s words in place (a resumed run continues the base-noise stream instead of replaying it)"""
    st = rng_state(device)
    st.copy_(torch.as_tensor(words, dtype=torch.int64).reshape(3))
    return st


def randn(rows, cols, device, scale=1.0, state=None):
    """[rows, cols] f32 ~ N(0, scale^2) drawn on the device by mhe_randn_f32 (graph-capturable: the launch advances its own counter)"""
    out = torch.empty(rows, cols, device=device, dtype=torch.float32)
    st = state if state is not None else rng_state(out.device)
    check(_lib.lib().mhe_randn_f32(_ptr(out), out.numel(), _ptr(st), float(scale), _stream()), "mhe_randn_f32")
    return out


def dropout_(x, p, bits=None, state=None):
    """x <- x * keep / (1 - p) in place (mhe_dropout).  bits=None: the mask is drawn on the device (the generator state of randn advances) and
    its bits are returned (uint8 [x.numel() / 8], bit k of byte i = element 8 i + k kept); bits given: that mask is applied."""
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_cuda or not x.is_contiguous() or x.numel() % 8:
        raise _lib.MheError("dropout_: contiguous f32 / bf16 device tensor with a multiple of 8 elements expected")
    draw = bits is None
    if draw:
        bits = torch.empty(x.numel() // 8, device=x.device, dtype=torch.uint8)
        st = state if state is not None else rng_state(x.device)
    else:
        _chk(bits, torch.uint8, "dropout.bits", (x.numel() // 8,))
        st = None
    check(_lib.lib().mhe_dropout(_ptr(x), dtype_code(x.dtype), _ptr(bits), x.numel(), float(p), _ptr(st), int(draw), _stream()), "mhe_dropout")
    return bits


def dropout_mask(bits, shape, p):
    """the float mask (0 or 1 / (1 - p)) that dropout_ applied, from its bits (tests: the oracle is fed the same mask)"""
    m = ((bits.view(-1, 1).int() >> torch.arange(8, device=bits.device, dtype=torch.int32)) & 1).float().reshape(shape)
    return m / (1.0 - p)


def reparam(mn, l2, eps=None, sigmoid_act=False, deterministic=False):
    """(sd, z) of BasicEnc's stochastic head (mhe_reparam_f32)"""
    _chk(mn, torch.float32, "reparam.mn"); _chk(l2, torch.float32, "reparam.l2", mn.shape)
    if eps is not None:
        _chk(eps, torch.float32, "reparam.eps", mn.shape)
    sd, z = torch.empty_like(mn), torch.empty_like(mn)
    check(_lib.lib().mhe_reparam_f32(_ptr(mn), _ptr(l2), _ptr(eps), _ptr(sd), _ptr(z), mn.numel(), int(sigmoid_act), int(deterministic or eps is None),
                                     _stream()), "mhe_reparam_f32")
    return sd, z


def flow_pack_net(w0, w1, w2):
    """Host repack of one coupling network's weights (numpy float32) into the
    fragment-ordered stream (mhe_flow_pack_net_host)."""
    hidden, dim = w0.shape
    L = _lib.lib()
    n = L.mhe_flow_packed_floats_per_net(dim, hidden)
    if n == 0:
        raise _lib.MheError(f"flow_pack_net: unsupported geometry dim={dim} hidden={hidden}")
    out = np.empty(n, np.float32)
    w0, w1, w2 = (np.ascontiguousarray(a, np.float32) for a in (w0, w1, w2))
    check(L.mhe_flow_pack_net_host(w0.ctypes.data_as(C.c_void_p), w1.ctypes.data_as(C.c_void_p),
                                   w2.ctypes.data_as(C.c_void_p), dim, hidden, out.ctypes.data_as(C.c_void_p)),
          "mhe_flow_pack_net_host")
    return out


def flow_pack_net_bf16(w0, w1, w2):
    """Host repack of one coupling network into the bf16 fragment stream (mhe_flow_pack_net_bf16_host)."""
    hidden, dim = w0.shape
    L = _lib.lib()
    n = L.mhe_flow_packed_bytes_per_net_bf16(dim, hidden)
    if n == 0:
        raise _lib.MheError(f"flow_pack_net_bf16: unsupported geometry dim={dim} hidden={hidden}")
    out = np.empty(n // 2, np.uint16)
    w0, w1, w2 = (np.ascontiguousarray(a, np.float32) for a in (w0, w1, w2))
    check(L.mhe_flow_pack_net_bf16_host(w0.ctypes.data_as(C.c_void_p), w1.ctypes.data_as(C.c_void_p),
                                        w2.ctypes.data_as(C.c_void_p), dim, hidden, out.ctypes.data_as(C.c_void_p)),
          "mhe_flow_pack_net_bf16_host")
    return out


def flow_couplings(x_in, cond, wstream, bias2, mask, B, hidden, direction, want_log_prob=True):
    """All couplings in one launch; returns (out, sum_s, log_prob).  The dtype of `wstream` selects the
    kernel: float32 stream -> f32 MFMA, int16/bf16 stream -> bf16 MFMA."""
    R, dim = x_in.shape
    ncoup = mask.shape[0]
    _chk(x_in, torch.float32, "flow.in")
    _chk(cond, torch.float32, "flow.cond", (B, 2 * ncoup, 2, hidden))
    bf16 = wstream.dtype != torch.float32
    _chk(wstream, wstream.dtype, "flow.wstream"); _chk(bias2, torch.float32, "flow.bias2", (2 * ncoup, 64 if bf16 else dim))
    _chk(mask, torch.float32, "flow.mask", (ncoup, dim))
    out = torch.empty_like(x_in)
    sum_s = torch.empty(R, device=x_in.device, dtype=torch.float32)
    logp = torch.empty(R, device=x_in.device, dtype=torch.float32) if want_log_prob else None
    fn = _lib.lib().mhe_flow_couplings_bf16 if bf16 else _lib.lib().mhe_flow_couplings_f32
    check(fn(_ptr(x_in), _ptr(out), _ptr(cond), _ptr(wstream), _ptr(bias2), _ptr(mask), _ptr(sum_s), _ptr(logp), R, B, dim,
             hidden, ncoup, direction, _stream()), "mhe_flow_couplings_bf16" if bf16 else "mhe_flow_couplings_f32")
    return out, sum_s, logp


def flow_couplings_emit(x_in, cond, wstream, bias2, mask, B, hidden, direction, h1, h2, o):
    """flow_couplings (bf16 stream, hidden 512) that also writes the nets' hidden activations h1, h2 (bf16 [nets, R, 512]) and
    the s / t pre-activations o (f32 [nets, R, 64]) for the reverse pass; returns (out, sum_s, log_prob)."""
    R, dim = x_in.shape
    ncoup = mask.shape[0]
    _chk(x_in, torch.float32, "flow.in"); _chk(cond, torch.float32, "flow.cond", (B, 2 * ncoup, 2, hidden))
    _chk(wstream, wstream.dtype, "flow.wstream"); _chk(bias2, torch.float32, "flow.bias2", (2 * ncoup, 64))
    _chk(mask, torch.float32, "flow.mask", (ncoup, dim))
    _chk(h1, torch.bfloat16, "flow.h1", (2 * ncoup, R, hidden)); _chk(h2, torch.bfloat16, "flow.h2", (2 * ncoup, R, hidden))
    _chk(o, torch.float32, "flow.o", (2 * ncoup, R, 64))
    out = torch.empty_like(x_in)
    sum_s = torch.empty(R, device=x_in.device, dtype=torch.float32)
    logp = torch.empty(R, device=x_in.device, dtype=torch.float32)
    check(_lib.lib().mhe_flow_couplings_bf16_emit(_ptr(x_in), _ptr(out), _ptr(cond), _ptr(wstream), _ptr(bias2), _ptr(mask), _ptr(sum_s),
                                                  _ptr(logp), _ptr(h1), _ptr(h2), _ptr(o), R, B, dim, hidden, ncoup, direction, _stream()),
          "mhe_flow_couplings_bf16_emit")
    return out, sum_s, logp


def flow_couplings_frag_supported(R, B, dim, hidden, ncoup):
    return bool(_lib.lib().mhe_flow_couplings_frag_supported(R, B, dim, hidden, ncoup))


def flow_frag_pack(w0, w1, w2):
    """one net's layer weights (torch [h, dim], [h, h], [dim, h]; or index tensors of those shapes) -> (w1F, w0F, w2F) in MFMA fragment order,
    dim zero-padded (index tensors: -1) to 64 - the operands of mhe_flow_couplings_frag_bf16"""
    h, dim = w0.shape
    fill = -1 if not w0.is_floating_point() else 0
    w0p = torch.full((h, 64), fill, dtype=w0.dtype, device=w0.device); w0p[:, :dim] = w0
    w2p = torch.full((64, h), fill, dtype=w2.dtype, device=w2.device); w2p[:dim] = w2
    return mfma_fragment_major(w1), mfma_fragment_major(w0p), mfma_fragment_major(w2p)


def flow_couplings_frag(x_in, cond, w0F, w1F, w2F, w_net_stride, bias2, mask, B, hidden, direction, want_log_prob=True, emit=None, sign_bits=None):
    """flow_couplings / flow_couplings_emit (emit = (h1, h2, o)) on the fragment-streaming kernel (mhe_flow_couplings_frag_bf16);
    sign_bits (with emit, int32 [nets, R / 64, 2, 8, 64, 2]): also the signs of h1 / h2 as the reverse chain reads them"""
    R, dim = x_in.shape
    ncoup = mask.shape[0]
    _chk(x_in, torch.float32, "flow.in"); _chk(cond, torch.float32, "flow.cond", (B, 2 * ncoup, 2, hidden))
    _chk(bias2, torch.float32, "flow.bias2", (2 * ncoup, 64)); _chk(mask, torch.float32, "flow.mask", (ncoup, dim))
    for t, nm in ((w0F, "w0F"), (w1F, "w1F"), (w2F, "w2F")):
        _chk(t, torch.bfloat16, "flow." + nm)
    h1 = h2 = o = None
    if emit is not None:
        h1, h2, o = emit
        _chk(h1, torch.bfloat16, "flow.h1", (2 * ncoup, R, hidden)); _chk(h2, torch.bfloat16, "flow.h2", (2 * ncoup, R, hidden))
        _chk(o, torch.float32, "flow.o", (2 * ncoup, R, 64))
    if sign_bits is not None:
        _chk(sign_bits, torch.int32, "flow.sign_bits", (2 * ncoup, R // 64, 2, 8, 64, 2))
    out = torch.empty_like(x_in)
    sum_s = torch.empty(R, device=x_in.device, dtype=torch.float32)
    logp = torch.empty(R, device=x_in.device, dtype=torch.float32) if want_log_prob else None
    check(_lib.lib().mhe_flow_couplings_frag_bf16(_ptr(x_in), _ptr(out), _ptr(cond), 4 * ncoup * hidden, _ptr(w0F), _ptr(w1F), _ptr(w2F),
                                                  int(w_net_stride), _ptr(bias2), _ptr(mask), _ptr(sum_s), _ptr(logp), _ptr(h1), _ptr(h2), _ptr(o),
                                                  _ptr(sign_bits), R, B, dim, hidden, ncoup, direction, _stream()), "mhe_flow_couplings_frag_bf16")
    return out, sum_s, logp


def mano_joints(th45, det, tables, crop_uv=None, vis=None, laplace_b=0.03, th45_alpha=50.0, inv_norm=False,
                image_size=256.0, want=("z", "xyz", "uv", "terms", "log_p", "norms")):
    """want may also include "joints_mm", and "verts" (the full mesh, normalised like xyz) or "mesh_mm" (ManoLayer's mesh in mm): the mesh
    of the same hypotheses through mhe_mano_decode_f32 - the joint pass leaves the skinning operands, no second pose pass."""
    R, B = th45.shape[0], det.shape[0]
    dev = th45.device
    _chk(th45, torch.float32, "mano.th45", (R, 45)); _chk(det, torch.float32, "mano.det", (B, 16))
    _chk(tables, torch.float32, "mano.tables")
    if crop_uv is not None:
        _chk(crop_uv, torch.float32, "mano.crop_uv", (B, 42)); _chk(vis, torch.float32, "mano.vis", (B, 21))
    shapes = {"z": (R, 61), "xyz": (R, 63), "uv": (R, 42), "terms": (R, 4), "log_p": (R,), "norms": (R, 2),
              "joints_mm": (R, 63)}
    o = {k: (torch.empty(shapes[k], device=dev, dtype=torch.float32) if k in want else None) for k in shapes}
    if "verts" in want or "mesh_mm" in want:
        if "verts" in want and "mesh_mm" in want:
            raise ValueError("mano_joints: 'verts' or 'mesh_mm', one mesh per call")
        key = "verts" if "verts" in want else "mesh_mm"
        o[key] = torch.empty(R, 778, 3, device=dev, dtype=torch.float32)
        ws = torch.empty(_lib.lib().mhe_mano_verts_workspace_floats(R), device=dev, dtype=torch.float32)
        check(_lib.lib().mhe_mano_decode_f32(_ptr(th45), _ptr(det), _ptr(crop_uv), _ptr(vis), _ptr(tables),
                                             _ptr(o["z"]), _ptr(o["xyz"]), _ptr(o["uv"]), _ptr(o["terms"]), _ptr(o["log_p"]),
                                             _ptr(o["norms"]), _ptr(o["joints_mm"]), _ptr(o[key]), _ptr(ws), R, B, float(laplace_b),
                                             float(th45_alpha), int(inv_norm), float(image_size), int(key == "mesh_mm"), _stream()),
              "mhe_mano_decode_f32")
        return o
    check(_lib.lib().mhe_mano_joints_f32(_ptr(th45), _ptr(det), _ptr(crop_uv), _ptr(vis), _ptr(tables),
                                         _ptr(o["z"]), _ptr(o["xyz"]), _ptr(o["uv"]), _ptr(o["terms"]), _ptr(o["log_p"]),
                                         _ptr(o["norms"]), _ptr(o["joints_mm"]), R, B, float(laplace_b), float(th45_alpha), int(inv_norm),
                                         float(image_size), _stream()), "mhe_mano_joints_f32")
    return o


def mano_joints_bwd(th45, det, tables, crop_uv, vis, g_log_p, N, laplace_b=0.03, th45_alpha=50.0):
    """reverse of mano_joints' log_p: (d/d th45 [R,45], d/d det [B,16]) for d loss/d log_p[n*B+b] = g_log_p[b]/N"""
    R, B = th45.shape[0], det.shape[0]
    _chk(th45, torch.float32, "mano_bwd.th45", (R, 45)); _chk(det, torch.float32, "mano_bwd.det", (B, 16))
    _chk(crop_uv, torch.float32, "mano_bwd.crop_uv", (B, 42)); _chk(vis, torch.float32, "mano_bwd.vis", (B, 21))
    _chk(g_log_p, torch.float32, "mano_bwd.g_log_p", (B,))
    g_th45 = torch.empty(R, 45, device=th45.device, dtype=torch.float32)
    g_rows = torch.empty(R, 16, device=th45.device, dtype=torch.float32)
    check(_lib.lib().mhe_mano_joints_bwd_f32(_ptr(th45), _ptr(det), _ptr(crop_uv), _ptr(vis), _ptr(tables), _ptr(g_log_p),
                                             _ptr(g_th45), _ptr(g_rows), R, B, float(laplace_b), float(th45_alpha), 1.0 / N,
                                             _stream()), "mhe_mano_joints_bwd_f32")
    return g_th45, sum_over_hypotheses(g_rows, N, B)


def sum_over_hypotheses(rows, N, B, out=None, accumulate=False, out_stride=0):
    """out[b] (+)= sum_n rows[n*B + b]; `out` may be a view into a wider [B, out_stride] matrix"""
    Cc = rows.shape[1]
    _chk(rows, torch.float32, "sum_over_hypotheses.rows", (N * B, Cc))
    if out is None:
        out = torch.empty(B, Cc, device=rows.device, dtype=torch.float32)
    check(_lib.lib().mhe_sum_over_hypotheses_f32(_ptr(rows), C.c_void_p(out.data_ptr()), N, B, Cc, int(accumulate), int(out_stride), _stream()),
          "mhe_sum_over_hypotheses_f32")
    return out


def mano_verts(z, tables, mm=False):
    R = z.shape[0]
    _chk(z, torch.float32, "mano_verts.z", (R, 61))
    verts = torch.empty(R, 778, 3, device=z.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().mhe_mano_verts_workspace_floats(R), device=z.device, dtype=torch.float32)
    check(_lib.lib().mhe_mano_verts_f32(_ptr(z), _ptr(tables), _ptr(verts), _ptr(ws), R, int(mm), _stream()), "mhe_mano_verts_f32")
    return verts


def mano_regress_joints(verts, tables):
    R = verts.shape[0]
    _chk(verts, torch.float32, "regress.verts", (R, 778, 3))
    j = torch.empty(R, 21, 3, device=verts.device, dtype=torch.float32)
    check(_lib.lib().mhe_mano_regress_joints_f32(_ptr(verts), _ptr(tables), _ptr(j), R, _stream()), "mhe_mano_regress_joints_f32")
    return j


def elbo_reduce(log_p_rows, log_q_rows, N, B):
    dev = log_p_rows.device
    _chk(log_p_rows, torch.float32, "elbo.log_p_rows", (N * B,))
    if log_q_rows is not None:
        _chk(log_q_rows, torch.float32, "elbo.log_q_rows", (N * B,))
    q, h, lp = (torch.empty(B, device=dev, dtype=torch.float32) for _ in range(3))
    check(_lib.lib().mhe_elbo_reduce_f32(_ptr(log_p_rows), _ptr(log_q_rows), _ptr(q), _ptr(h), _ptr(lp), N, B, _stream()),
          "mhe_elbo_reduce_f32")
    return q, h, lp


def topk_gather(score, rows, N, B, Q):
    """per image the Q rows of highest score, descending (torch.topk order): returns (idx [Q,B] int32, rows [Q*B,D])"""
    D = rows.shape[1]
    _chk(score, torch.float32, "topk.score", (N * B,)); _chk(rows, torch.float32, "topk.rows", (N * B, D))
    idx = torch.empty(Q, B, device=rows.device, dtype=torch.int32)
    out = torch.empty(Q * B, D, device=rows.device, dtype=torch.float32)
    check(_lib.lib().mhe_topk_gather_f32(_ptr(score), _ptr(rows), _ptr(idx), _ptr(out), N, B, Q, D, _stream()),
          "mhe_topk_gather_f32")
    return idx, out


def metrics(xyz, uv, pose3d, scale, crop_uv, vis):
    N, B = xyz.shape[:2]
    _chk(xyz, torch.float32, "metrics.xyz", (N, B, 63)); _chk(uv, torch.float32, "metrics.uv", (N, B, 42))
    _chk(pose3d, torch.float32, "metrics.pose3d", (B, 63)); _chk(scale, torch.float32, "metrics.scale", (B,))
    _chk(crop_uv, torch.float32, "metrics.crop_uv", (B, 42)); _chk(vis, torch.float32, "metrics.vis", (B, 21))
    out = torch.empty(14, B, device=xyz.device, dtype=torch.float32)
    check(_lib.lib().mhe_metrics_f32(_ptr(xyz), _ptr(uv), _ptr(pose3d), _ptr(scale), _ptr(crop_uv), _ptr(vis), _ptr(out),
                                     N, B, _stream()), "mhe_metrics_f32")
    return out


def conv2d_nhwc(x, w, KH, KW, stride, pad, in_scale=None, in_shift=None, relu_in=False, out_scale=None,
                out_shift=None, residual=None, relu_out=False, stats=None, out=None, mask=None, bn=None, tile=0, res_half=False, xcat=None,
                mask_bits=None):
    """x [B,H,W,Cin], w packed [Cout, Kpad]; returns y [B,Ho,Wo,Cout] of x.dtype.  mask (shaped like y, data-gradient
    form only): y = (conv + residual) * [mask > 0]; bn = up to two (bn_y, mean_invstd [2,C], stats [S,2,C]) triples: the epilogue
    also accumulates the BatchNorm-reverse sums of y for those units (see mhe_conv2d_masked_nhwc).  tile > 0 forces kernel
    variant tile-1 (mhe_conv_desc.tile; tests and tuning)."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    dt = x.dtype
    _chk(x, dt, "conv.x"); _chk(w, dt, "conv.w")
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    y = out if out is not None else torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=dt)
    _chk(y, dt, "conv.y", (B, Ho, Wo, Cout))
    for t, n, c in ((in_scale, "in_scale", Cin), (in_shift, "in_shift", Cin), (out_scale, "out_scale", Cout),
                    (out_shift, "out_shift", Cout)):
        if t is not None:
            _chk(t, torch.float32, "conv." + n, (c,))
    if residual is not None:
        _chk(residual, dt, "conv.residual", (B, (Ho + 1) // 2, (Wo + 1) // 2, Cout) if res_half else (B, Ho, Wo, Cout))
    if stats is not None:
        _chk_stats(stats, "conv.stats", Cout)
    d = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride, pad, dtype_code(dt), int(relu_in), int(relu_out), int(tile), int(res_half))
    if mask is not None:
        if in_scale is not None or out_scale is not None or stats is not None or relu_in or relu_out:
            raise ValueError("conv2d_nhwc: mask= is the plain data-gradient form (no affine / statistics / relu; out_shift = a per-channel constant)")
        _chk(mask, dt, "conv.mask", (B, Ho, Wo, Cout))
        if out_shift is not None:        # y = (conv + out_shift + residual) [mask > 0], one consumer's sums at most
            if bn is not None and len(bn) > 1:
                raise ValueError("conv2d_nhwc: out_shift= with mask= takes one bn consumer")
            by, bmi, bst = (list(bn or []) + [(None, None, None)])[0]
            if by is not None:
                _chk(by, dt, "conv.bn_y", (B, Ho, Wo, Cout)); _chk(bmi, torch.float32, "conv.bn_mean_invstd", (2, Cout))
                _chk_stats(bst, "conv.bn_stats", Cout)
            cin2 = 0
            if xcat is not None:         # the operand's K range continued on a second tensor: w is [Cout][Cin + cin2]
                cin2 = xcat.shape[-1]
                _chk(xcat, dt, "conv.xcat", (B, H, W, cin2))
                if tuple(w.shape) != (Cout, Cin + cin2):
                    raise ValueError(f"conv2d_nhwc: xcat= needs w [{Cout}, {Cin + cin2}], got {tuple(w.shape)}")
            with _dg_timed(d, x, y, w, residual, mask, None, [by], "cat" if xcat is not None else "bias", xcat):
                check(_lib.lib().mhe_conv2d_masked_bias_nhwc(C.byref(d), _ptr(x), _ptr(xcat), int(cin2), _ptr(w), _ptr(y), _ptr(residual), _ptr(mask),
                                                             _ptr(out_shift), _ptr(by), _ptr(bmi), _ptr(bst), _stream()), "mhe_conv2d_masked_bias_nhwc")
            return y
        if xcat is not None:
            raise ValueError("conv2d_nhwc: xcat= comes with mask= and out_shift=")
        ext = []
        for by, bmi, bst in (list(bn or []) + [(None, None, None)] * 2)[:2]:
            if by is not None:
                _chk(by, dt, "conv.bn_y", (B, Ho, Wo, Cout)); _chk(bmi, torch.float32, "conv.bn_mean_invstd", (2, Cout))
                _chk_stats(bst, "conv.bn_stats", Cout)
            ext += [_ptr(by), _ptr(bmi), _ptr(bst)]
        if mask_bits is not None:        # the gate also as bits [pixel][Cout / 8] (bottleneck_tail want_bits=): read instead of `mask` where the kernel can
            _chk(mask_bits, torch.uint8, "conv.mask_bits", (B, Ho, Wo, Cout // 8))
            with _dg_timed(d, x, y, w, residual, mask, mask_bits, [b[0] for b in (bn or [])], "bits"):
                check(_lib.lib().mhe_conv2d_masked_bits_nhwc(C.byref(d), _ptr(x), _ptr(w), _ptr(y), _ptr(residual), _ptr(mask), _ptr(mask_bits), *ext, _stream()),
                      "mhe_conv2d_masked_bits_nhwc")
            return y
        with _dg_timed(d, x, y, w, residual, mask, None, [b[0] for b in (bn or [])], ""):
            check(_lib.lib().mhe_conv2d_masked_nhwc(C.byref(d), _ptr(x), _ptr(w), _ptr(y), _ptr(residual), _ptr(mask), *ext, _stream()),
                  "mhe_conv2d_masked_nhwc")
        return y
    if bn:
        raise ValueError("conv2d_nhwc: bn= needs mask= (data-gradient form)")
    if xcat is not None:                 # ungated product on two operand tensors (+ out_shift as a per-channel constant, + residual)
        if in_scale is not None or out_scale is not None or stats is not None or relu_in or relu_out:
            raise ValueError("conv2d_nhwc: xcat= without mask= takes out_shift and residual only")
        cin2 = xcat.shape[-1]
        _chk(xcat, dt, "conv.xcat", (B, H, W, cin2))
        if tuple(w.shape) != (Cout, Cin + cin2):
            raise ValueError(f"conv2d_nhwc: xcat= needs w [{Cout}, {Cin + cin2}], got {tuple(w.shape)}")
        check(_lib.lib().mhe_conv1x1_cat_bias_nhwc(C.byref(d), _ptr(x), _ptr(xcat), int(cin2), _ptr(w), _ptr(y), _ptr(residual), _ptr(out_shift), _stream()),
              "mhe_conv1x1_cat_bias_nhwc")
        return y
    if TIMING:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(_lib.lib().mhe_conv2d_nhwc(C.byref(d), _ptr(x), _ptr(w), _ptr(y), _ptr(in_scale), _ptr(in_shift),
                                     _ptr(out_scale), _ptr(out_shift), _ptr(residual), _ptr(stats), _stream()),
          "mhe_conv2d_nhwc")
    if TIMING:
        ev1.record()
        es = x.element_size()      # algorithmic HBM bytes: input once, output once, weights once (+ residual)
        nbytes = es * (x.numel() + y.numel() + w.numel() + (residual.numel() if residual is not None else 0))
        KERNEL_TIMES.append((_conv_kernel_name(d, dt, 1 if in_scale is not None else 3 if residual is not None else 0),
                             2.0 * B * Ho * Wo * Cout * KH * KW * Cin, ev0, ev1, nbytes))
    return y


def conv1x1_stats(x, w, in_scale, in_shift, stats):
    """batch statistics of conv1x1(relu(x * in_scale + in_shift), w) as it would be stored, without storing it (mhe_conv1x1_stats_nhwc)"""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    _chk(x, torch.bfloat16, "conv_stats.x"); _chk(w, torch.bfloat16, "conv_stats.w", (Cout, Cin))
    _chk(in_scale, torch.float32, "conv_stats.in_scale", (Cin,)); _chk(in_shift, torch.float32, "conv_stats.in_shift", (Cin,))
    _chk_stats(stats, "conv_stats.stats", Cout)
    d = ConvDesc(B, H, W, Cin, Cout, 1, 1, 1, 0, BF16, 1, 0, 0, 0)
    if TIMING:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(_lib.lib().mhe_conv1x1_stats_nhwc(C.byref(d), _ptr(x), _ptr(w), _ptr(in_scale), _ptr(in_shift), _ptr(stats), _stream()), "mhe_conv1x1_stats_nhwc")
    if TIMING:
        ev1.record()
        name = "mhe::conv::conv_wide_kernel<128, 4, true>" if Cin == 256 else "mhe::conv::conv1x1_stream_kernel<%d, 256, true, false>" % (Cin // 64)
        KERNEL_TIMES.append((name, 2.0 * B * H * W * Cout * Cin, ev0, ev1, 2 * (x.numel() + w.numel())))
    return stats


def gram_buffers(Cb, device):
    """(partial-slab workspace, f64 totals) for conv1x1_gram_bn: every workgroup of the Gram launch stores its partial sums as a slab, the
    finalize sums the slabs in slab order (csrc/conv_gram.hip) - nothing to clear between steps"""
    L = _lib.lib()
    return (torch.zeros(L.mhe_gram_stats_words(Cb), device=device, dtype=STAT_DTYPE),
            torch.empty(L.mhe_gram_stats_workspace_bytes(Cb) // 8, device=device, dtype=torch.float64))


def conv1x1_gram_bn(x, in_scale, in_shift, w, bn_weight, bn_bias, running_mean, running_var, bufs, momentum=0.1, eps=1e-5, num_batches_tracked=None,
                    want_mean_invstd=False, a_out=None):
    """train-mode BatchNorm affine (scale, shift) of conv1x1(relu(x * in_scale + in_shift), w) from the Gram matrix of the convolution's
    INPUT - the product itself is never evaluated (mhe_conv1x1_gram_nhwc + mhe_gram_bn_finalize; csrc/conv_gram.hip)"""
    B, H, W, Cb = x.shape
    Cn = w.shape[0]
    _chk(x, torch.bfloat16, "gram.x"); _chk(w, torch.bfloat16, "gram.w", (Cn, Cb))
    _chk(in_scale, torch.float32, "gram.in_scale", (Cb,)); _chk(in_shift, torch.float32, "gram.in_shift", (Cb,))
    gram, ws = bufs
    L = _lib.lib()
    _chk(gram, STAT_DTYPE, "gram.accumulators", (L.mhe_gram_stats_words(Cb),))
    if TIMING:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if a_out is not None:        # also relu(x * in_scale + in_shift) itself, as the matrix cores multiplied it
        _chk(a_out, torch.bfloat16, "gram.a_out", (B, H, W, Cb))
    check(L.mhe_conv1x1_gram_store_nhwc(_ptr(x), _ptr(in_scale), _ptr(in_shift), 1, _ptr(gram), _ptr(a_out), B * H * W, Cb, _stream()), "mhe_conv1x1_gram_store_nhwc")
    if TIMING:
        ev1.record()
        KERNEL_TIMES.append(("mhe::conv::gram_kernel<%d>" % Cb, 2.0 * B * H * W * Cb * Cb, ev0, ev1, 2 * x.numel()))
    scale = torch.empty(Cn, device=x.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    mi = torch.empty(2, Cn, device=x.device, dtype=torch.float32) if want_mean_invstd else None
    check(L.mhe_gram_bn_finalize(_ptr(gram), _ptr(ws), _ptr(w), _ptr(bn_weight), _ptr(bn_bias), _ptr(running_mean), _ptr(running_var), _ptr(scale),
                                 _ptr(shift), _ptr(mi), Cn, Cb, float(B * H * W), float(momentum), float(eps), _ptr(num_batches_tracked), _stream()),
          "mhe_gram_bn_finalize")
    return (scale, shift, mi) if want_mean_invstd else (scale, shift)


def gram_workspace(Cb, device):
    """an f64 workspace of conv1x1_gram_bn of one's own (its totals - Gram matrix [Cb][Cb], then column sums [Cb] - stay valid until reused)"""
    return torch.empty(_lib.lib().mhe_gram_stats_workspace_bytes(Cb) // 8, device=device, dtype=torch.float64)


def conv3_bn_fold(D, w, gram_totals, rev_stats, gamma, mean_invstd, count, dgamma, dbeta, dW, w_dg, S, c0, coef_ws):
    """reverse of conv3 + train-mode BatchNorm from D = g^T A and the forward's Gram statistics (mhe_conv3_bn_fold; csrc/conv_fold.hip).
    S=None: w_dg is [Cb][C + Cb] and receives [(k2 W)^T | S] - the weights of ONE K-concatenated data-gradient launch (conv2d_nhwc xcat=)"""
    Cn, Cb = w.shape
    if S is None:
        _chk(w_dg, torch.bfloat16, "fold.w_cat", (Cb, Cn + Cb))
        s_ptr, ld_s = C.c_void_p(w_dg.data_ptr() + 2 * Cn), Cn + Cb
    else:
        _chk(S, torch.bfloat16, "fold.S", (Cb, Cb))
        s_ptr, ld_s = _ptr(S), Cb
    _chk(D, torch.float32, "fold.D", (Cn, Cb)); _chk(w, torch.bfloat16, "fold.w", (Cn, Cb)); _chk(gram_totals, torch.float64, "fold.gram")
    _chk_stats(rev_stats, "fold.rev_stats", Cn); _chk(gamma, torch.float32, "fold.gamma", (Cn,))
    _chk(mean_invstd, torch.float32, "fold.mean_invstd", (2, Cn)); _chk(dgamma, torch.float32, "fold.dgamma", (Cn,)); _chk(dbeta, torch.float32, "fold.dbeta", (Cn,))
    _chk(dW, torch.float32, "fold.dW"); _chk(w_dg, torch.bfloat16, "fold.w_dg")
    _chk(c0, torch.float32, "fold.c0", (Cb,)); _chk(coef_ws, torch.float32, "fold.coef_ws")
    if gram_totals.numel() < Cb * Cb + Cb or dW.numel() < Cn * Cb or w_dg.shape[0] != Cb or w_dg.shape[1] < Cn or coef_ws.numel() < 2 * Cn:
        raise ValueError("conv3_bn_fold: operand sizes")
    check(_lib.lib().mhe_conv3_bn_fold(_ptr(D), _ptr(w), _ptr(gram_totals), _ptr(rev_stats), _ptr(gamma), _ptr(mean_invstd), float(count), _ptr(dgamma),
                                       _ptr(dbeta), _ptr(dW), _ptr(w_dg), int(w_dg.shape[1]), s_ptr, int(ld_s), _ptr(c0), _ptr(coef_ws), Cn, Cb, _stream()),
          "mhe_conv3_bn_fold")


def conv3x3_halo_supported(B, H, W, Cin, Cout):
    return bool(_lib.lib().mhe_conv3x3_halo_supported(int(B), int(H), int(W), int(Cin), int(Cout)))


def conv3x3_halo_pack(w):
    """standard bf16 pack [Cout][9 Cin] -> the fragment-major 16 KiB stages csrc/conv_halo.hip streams (same byte count)"""
    _chk(w, torch.bfloat16, "halo_pack.w")
    Cout, K = w.shape
    if K % 9 or (K // 9) % 64 or Cout % 128:
        raise ValueError(f"conv3x3_halo_pack: weight rows [{Cout}][{K}] are not 9 x (a multiple of 64) wide / a multiple of 128 many")
    out = torch.empty_like(w)
    check(_lib.lib().mhe_conv3x3_halo_pack_bf16(_ptr(w), _ptr(out), int(Cout), int(K // 9), _stream()), "mhe_conv3x3_halo_pack_bf16")
    return out


def conv3x3_halo(x, w_halo, in_scale=None, in_shift=None, relu_in=False, a_out=None, stats=None, residual=None, mask=None, bn=None):
    """3x3 / stride 1 / pad 1 with the input tile resident in LDS (mhe_conv3x3_halo_nhwc).  in_scale / in_shift: the producer's BatchNorm on
    the way in; a_out: that operand written once; mask (+ residual, bn = (raw output, mean_invstd, stats) of ONE consumer): data-gradient form."""
    B, H, W, Cin = x.shape
    Cout = w_halo.shape[0]
    _chk(x, torch.bfloat16, "halo.x"); _chk(w_halo, torch.bfloat16, "halo.w", (Cout, 9 * Cin))
    y = torch.empty(B, H, W, Cout, device=x.device, dtype=torch.bfloat16)
    for t, name, shape in ((in_scale, "in_scale", (Cin,)), (in_shift, "in_shift", (Cin,))):
        if t is not None:
            _chk(t, torch.float32, "halo." + name, shape)
    if stats is not None:
        _chk_stats(stats, "halo.stats", Cout)
    for t, name, shape in ((a_out, "a_out", tuple(x.shape)), (residual, "residual", tuple(y.shape)), (mask, "mask", tuple(y.shape))):
        if t is not None:
            _chk(t, torch.bfloat16, "halo." + name, shape)
    by, bmi, bst = (None, None, None) if bn is None else bn
    if by is not None:
        _chk(by, torch.bfloat16, "halo.bn_y", tuple(y.shape)); _chk(bmi, torch.float32, "halo.bn_mi", (2, Cout)); _chk_stats(bst, "halo.bn_stats", Cout)
    if TIMING:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(_lib.lib().mhe_conv3x3_halo_nhwc(B, H, W, Cin, Cout, _ptr(x), _ptr(w_halo), _ptr(y), _ptr(in_scale), _ptr(in_shift), int(bool(relu_in)),
                                           _ptr(a_out), _ptr(stats), _ptr(residual), _ptr(mask), _ptr(by), _ptr(bmi), _ptr(bst), _stream()),
          "mhe_conv3x3_halo_nhwc")
    if TIMING:
        ev1.record()
        KERNEL_TIMES.append(("mhe::conv::conv_halo_kernel<%d, %s>" % (W, "true" if mask is not None else "false"), 2.0 * B * H * W * Cout * 9 * Cin, ev0, ev1,
                             2 * (x.numel() + y.numel() + w_halo.numel())))
    return y


def conv3x3_halo_dgrad_bn(g, y_raw, coef, w_halo, mask, gy_out=None, residual=None, bn=None):
    """data gradient of a 3x3 / stride-1 unit with the BatchNorm reverse of its own output gradient on the operand load: operand = k2 g + k1 y_raw
    + k0 (coef [3, C] of bn_backward(coef_only=True)), written to gy_out for the weight gradient (mhe_conv3x3_halo_dgrad_bn_nhwc)"""
    B, H, W, Cin = g.shape
    Cout = w_halo.shape[0]
    _chk(g, torch.bfloat16, "halo_dg.g"); _chk(y_raw, torch.bfloat16, "halo_dg.y", tuple(g.shape)); _chk(coef, torch.float32, "halo_dg.coef", (3, Cin))
    _chk(w_halo, torch.bfloat16, "halo_dg.w", (Cout, 9 * Cin)); _chk(mask, torch.bfloat16, "halo_dg.mask", (B, H, W, Cout))
    gx = torch.empty(B, H, W, Cout, device=g.device, dtype=torch.bfloat16)
    if gy_out is not None:
        _chk(gy_out, torch.bfloat16, "halo_dg.gy_out", tuple(g.shape))
    if residual is not None:
        _chk(residual, torch.bfloat16, "halo_dg.residual", tuple(gx.shape))
    by, bmi, bst = (None, None, None) if bn is None else bn
    if by is not None:
        _chk(by, torch.bfloat16, "halo_dg.bn_y", tuple(gx.shape)); _chk(bmi, torch.float32, "halo_dg.bn_mi", (2, Cout)); _chk_stats(bst, "halo_dg.bn_stats", Cout)
    check(_lib.lib().mhe_conv3x3_halo_dgrad_bn_nhwc(B, H, W, Cin, Cout, _ptr(g), _ptr(y_raw), _ptr(coef), _ptr(w_halo), _ptr(gx), _ptr(gy_out), _ptr(residual),
                                                    _ptr(mask), _ptr(by), _ptr(bmi), _ptr(bst), _stream()), "mhe_conv3x3_halo_dgrad_bn_nhwc")
    return gx


def bottleneck_tail_supported(B, H, W, Cb, Cout):
    d = ConvDesc(B, H, W, 4 * Cb, Cout, 1, 1, 1, 0, BF16, 1, 0, 0, 0)
    return bool(_lib.lib().mhe_bottleneck_tail_supported(C.byref(d), int(Cb)))


def bottleneck_tail(y2, bn2, w3, bn3, identity, id_aff, w1, stats=None, want_bits=False):
    """(a, y1) of mhe_bottleneck_tail_nhwc: a = relu(bn3(conv3(relu(bn2(y2)))) + identity) with conv3 re-evaluated in place of being read
    back, y1 = the next block's conv1 of a (+ its batch statistics).  bn2 / bn3 / id_aff = (scale, shift) pairs (id_aff may be None)."""
    B, H, W, Cb = y2.shape
    Cw, Cout = w3.shape[0], w1.shape[0]
    _chk(y2, torch.bfloat16, "tail.y2"); _chk(w3, torch.bfloat16, "tail.w3", (Cw, Cb)); _chk(w1, torch.bfloat16, "tail.w1", (Cout, Cw))
    _chk(identity, torch.bfloat16, "tail.identity", (B, H, W, Cw))
    for (sc, sh), n, c in ((bn2, "bn2", Cb), (bn3, "bn3", Cw)) + (((id_aff, "id", Cw),) if id_aff is not None else ()):
        _chk(sc, torch.float32, f"tail.{n}_scale", (c,)); _chk(sh, torch.float32, f"tail.{n}_shift", (c,))
    if stats is not None:
        _chk_stats(stats, "tail.stats", Cout)
    a = torch.empty(B, H, W, Cw, device=y2.device, dtype=torch.bfloat16)
    y1 = torch.empty(B, H, W, Cout, device=y2.device, dtype=torch.bfloat16)
    # want_bits: also [a > 0] as bits, byte [pixel][channel / 8] - the gate the reverse pass reads instead of `a` (conv2d_nhwc mask_bits=)
    bits = torch.empty(B, H, W, Cw // 8, device=y2.device, dtype=torch.uint8) if want_bits else None
    d = ConvDesc(B, H, W, Cw, Cout, 1, 1, 1, 0, BF16, 1, 0, 0, 0)
    if TIMING:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(_lib.lib().mhe_bottleneck_tail_bits_nhwc(C.byref(d), Cb, _ptr(y2), _ptr(bn2[0]), _ptr(bn2[1]), _ptr(w3), _ptr(bn3[0]), _ptr(bn3[1]), _ptr(identity),
                                                   _ptr(id_aff[0] if id_aff is not None else None), _ptr(id_aff[1] if id_aff is not None else None),
                                                   _ptr(w1), _ptr(a), _ptr(bits), _ptr(y1), _ptr(stats), _stream()), "mhe_bottleneck_tail_bits_nhwc")
    if TIMING:
        ev1.record()
        nbytes = 2 * (y2.numel() + identity.numel() + a.numel() + y1.numel() + w3.numel() + w1.numel())
        KERNEL_TIMES.append(("mhe::conv::bottleneck_tail_kernel<%d, %d>" % (Cb, Cout), 2.0 * B * H * W * Cw * (Cb + Cout), ev0, ev1, nbytes))
    return (a, y1, bits) if want_bits else (a, y1)


def linear_bf16_f32out(x, w, bias=None, out=None):
    """out[R,N] (f32) = x[R,K] (bf16) w[N,K]^T (bf16) + bias, f32 accumulation (mhe_conv2d_f32out_nhwc); K % 64 == 0, N % 4 == 0"""
    R, K = x.shape
    N = w.shape[0]
    _chk(x, torch.bfloat16, "linear_bf16.x"); _chk(w, torch.bfloat16, "linear_bf16.w", (N, K))
    if bias is not None:
        _chk(bias, torch.float32, "linear_bf16.bias", (N,))
    y = out if out is not None else torch.empty(R, N, device=x.device, dtype=torch.float32)
    _chk(y, torch.float32, "linear_bf16.out", (R, N))
    d = ConvDesc(R, 1, 1, K, N, 1, 1, 1, 0, BF16, 0, 0, 0)
    check(_lib.lib().mhe_conv2d_f32out_nhwc(C.byref(d), _ptr(x), _ptr(w), _ptr(y), _ptr(bias), _stream()), "mhe_conv2d_f32out_nhwc")
    return y


def conv3x3s2_dgrad(gy, w4, residual=None, mask=None, bn=None, tile=0):
    """dx [B,2Ho,2Wo,Cin] of a 3x3 / stride-2 / pad-1 convolution from gy [B,Ho,Wo,Cout] by the four parity classes
    (mhe_conv3x3s2_dgrad_nhwc); w4 = the four packed tap subsets (train.dgrad_s2_operand_indices)."""
    B, Ho, Wo, Cout = gy.shape
    dt = gy.dtype
    Cin = w4[0].shape[0]
    _chk(gy, dt, "dgrad_s2.gy")
    for i, w in enumerate(w4):
        _chk(w, dt, f"dgrad_s2.w{i}")
    dx = torch.empty(B, 2 * Ho, 2 * Wo, Cin, device=gy.device, dtype=dt)
    for t, name in ((residual, "residual"), (mask, "mask")):
        if t is not None:
            _chk(t, dt, "dgrad_s2." + name, dx.shape)
    ext = []
    for by, bmi, bst in (list(bn or []) + [(None, None, None)] * 2)[:2]:
        if by is not None:
            _chk(by, dt, "dgrad_s2.bn_y", dx.shape); _chk(bmi, torch.float32, "dgrad_s2.bn_mean_invstd", (2, Cin))
            _chk_stats(bst, "dgrad_s2.bn_stats", Cin)
        ext += [_ptr(by), _ptr(bmi), _ptr(bst)]
    wp = (C.c_void_p * 4)(*[w.data_ptr() for w in w4])
    check(_lib.lib().mhe_conv3x3s2_dgrad_nhwc(B, Ho, Wo, Cout, Cin, dtype_code(dt), _ptr(gy), wp, _ptr(dx), _ptr(residual), _ptr(mask), *ext, int(tile), _stream()),
          "mhe_conv3x3s2_dgrad_nhwc")
    return dx


def conv1x1_residual_in(x, x2, w, in_scale, in_shift, x2_scale=None, x2_shift=None, a_out=None, stats=None, tile=0):
    """y = conv1x1(relu(x*in_scale+in_shift + (x2*x2_scale+x2_shift | x2))); optionally writes that operand to a_out."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    dt = x.dtype
    _chk(x, dt, "conv_res.x"); _chk(x2, dt, "conv_res.x2", x.shape); _chk(w, dt, "conv_res.w")
    _chk(in_scale, torch.float32, "conv_res.in_scale", (Cin,)); _chk(in_shift, torch.float32, "conv_res.in_shift", (Cin,))
    if x2_scale is not None:
        _chk(x2_scale, torch.float32, "conv_res.x2_scale", (Cin,)); _chk(x2_shift, torch.float32, "conv_res.x2_shift", (Cin,))
    if a_out is not None:
        _chk(a_out, dt, "conv_res.a_out", x.shape)
    if stats is not None:
        _chk_stats(stats, "conv_res.stats", Cout)
    y = torch.empty(B, H, W, Cout, device=x.device, dtype=dt)
    d = ConvDesc(B, H, W, Cin, Cout, 1, 1, 1, 0, dtype_code(dt), 1, 0, int(tile))
    if TIMING:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(_lib.lib().mhe_conv1x1_residual_in_nhwc(C.byref(d), _ptr(x), _ptr(x2), _ptr(w), _ptr(y), _ptr(in_scale), _ptr(in_shift),
                                                  _ptr(x2_scale), _ptr(x2_shift), _ptr(a_out), _ptr(stats), _stream()),
          "mhe_conv1x1_residual_in_nhwc")
    if TIMING:
        ev1.record()
        es = x.element_size()
        nbytes = es * (2 * x.numel() + y.numel() + w.numel() + (a_out.numel() if a_out is not None else 0))
        name = _conv_kernel_name(d, dt, 2)
        KERNEL_TIMES.append((name, 2.0 * B * H * W * Cout * Cin, ev0, ev1, nbytes))
    return y


def conv1x1_dgrad_bn_apply(g, y_raw, coef, w_dg, a_out, mask, bn=None, zeros=None, tile=0):
    """data gradient of a 1x1 convolution whose output gradient is still to receive its BatchNorm reverse: the operand
    gy = k2 g + k1 y_raw + k0 (coef = [k2, k1, k0] of mhe_bn_bwd_finalize) is formed in the operand load and written once to a_out;
    the result is gated by mask and (optionally) feeds the BatchNorm-reverse sums of one consumer (mhe_conv1x1_residual_in_masked_nhwc)."""
    B, H, W, Cin = g.shape
    Cout = w_dg.shape[0]
    dt = g.dtype
    _chk(g, dt, "dgrad_apply.g"); _chk(y_raw, dt, "dgrad_apply.y", g.shape); _chk(w_dg, dt, "dgrad_apply.w"); _chk(a_out, dt, "dgrad_apply.a_out", g.shape)
    _chk(coef, torch.float32, "dgrad_apply.coef", (3, Cin)); _chk(mask, dt, "dgrad_apply.mask", (B, H, W, Cout))
    if zeros is None:
        zeros = torch.zeros(Cin, device=g.device, dtype=torch.float32)
    _chk(zeros, torch.float32, "dgrad_apply.zeros", (Cin,))
    out = torch.empty(B, H, W, Cout, device=g.device, dtype=dt)
    by, bmi, bst = bn[0] if bn else (None, None, None)
    if by is not None:
        _chk(by, dt, "dgrad_apply.bn_y", out.shape); _chk(bmi, torch.float32, "dgrad_apply.bn_mean_invstd", (2, Cout))
        _chk_stats(bst, "dgrad_apply.bn_stats", Cout)
    d = ConvDesc(B, H, W, Cin, Cout, 1, 1, 1, 0, dtype_code(dt), 0, 0, int(tile))
    check(_lib.lib().mhe_conv1x1_residual_in_masked_nhwc(C.byref(d), _ptr(g), _ptr(y_raw), _ptr(w_dg), _ptr(out), _ptr(coef[0]), _ptr(coef[2]),
                                                         _ptr(coef[1]), _ptr(zeros), _ptr(a_out), None, _ptr(mask), _ptr(by), _ptr(bmi), _ptr(bst),
                                                         _stream()), "mhe_conv1x1_residual_in_masked_nhwc")
    return out


def stem_conv7x7s2(x, w, dtype, stats=None):
    """x [B,3,H,W] f32 NCHW, w packed [64, Kpad] (dtype) -> raw conv1 output [B,Ho,Wo,64] NHWC (dtype)."""
    B, Cn, H, W = x.shape
    _chk(x, torch.float32, "stem.x"); _chk(w, dtype, "stem.w", (64, 192))
    if Cn != 3:
        raise _lib.MheError("stem.x: expected 3 input channels")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, 64, device=x.device, dtype=dtype)
    if stats is not None:
        _chk_stats(stats, "stem.stats", 64)
    if TIMING:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(_lib.lib().mhe_stem_conv7x7s2(_ptr(x), _ptr(w), _ptr(y), _ptr(stats), B, H, W, dtype_code(dtype), _stream()),
          "mhe_stem_conv7x7s2")
    if TIMING:
        ev1.record()
        KERNEL_TIMES.append(("mhe::conv::stem_kernel<%s>" % ("float" if dtype == torch.float32 else "unsigned short"),
                             2.0 * B * Ho * Wo * 64 * 147, ev0, ev1, 4 * x.numel() + y.element_size() * y.numel()))
    return y


def stem_pool_supported(B, H, W, dtype):
    return bool(_lib.lib().mhe_stem_pool_supported(B, H, W, dtype_code(dtype)))


def stem_conv7x7s2_pool(x, w, bn_gamma, stats=None):
    """x [B,3,256,256] f32 NCHW -> pooled [B,64,64,64] bf16: per channel the 3x3 / stride-2 window max (gamma >= 0) or min (gamma < 0) of the raw
    conv1 output; relu(scale * pooled + shift) = maxpool(relu(bn1(conv1(x)))) (mhe_stem_conv7x7s2_pool).  stats: conv1's batch statistics."""
    B, Cn, H, W = x.shape
    _chk(x, torch.float32, "stem_pool.x"); _chk(w, torch.bfloat16, "stem_pool.w", (64, 192)); _chk(bn_gamma, torch.float32, "stem_pool.gamma", (64,))
    if Cn != 3:
        raise _lib.MheError("stem_pool.x: expected 3 input channels")
    if stats is not None:
        _chk_stats(stats, "stem_pool.stats", 64)
    y = torch.empty(B, 64, 64, 64, device=x.device, dtype=torch.bfloat16)
    if TIMING:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(_lib.lib().mhe_stem_conv7x7s2_pool(_ptr(x), _ptr(w), _ptr(bn_gamma), _ptr(y), _ptr(stats), B, H, W, _stream()), "mhe_stem_conv7x7s2_pool")
    if TIMING:
        ev1.record()
        KERNEL_TIMES.append(("mhe::conv::stem_pool_kernel", 2.0 * B * 128 * 128 * 64 * 147, ev0, ev1, 4 * x.numel() + 2 * y.numel()))
    return y


def bn_finalize(stats, gamma, beta, running_mean, running_var, count, momentum=0.1, eps=1e-5, want_mean_invstd=False, clear=False,
                num_batches_tracked=None):
    """clear: the accumulators are zeroed once read (self-cleaning arena); num_batches_tracked (int64 scalar on the device): += 1"""
    Cn = gamma.shape[0]
    _chk_stats(stats, "bn_finalize.stats", Cn)
    scale = torch.empty(Cn, device=gamma.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    mi = torch.empty(2, Cn, device=gamma.device, dtype=torch.float32) if want_mean_invstd else None
    if clear or num_batches_tracked is not None:
        if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
            raise _lib.MheError("bn_finalize.num_batches_tracked: int64 device tensor expected")
        check(_lib.lib().mhe_bn_finalize_step(_ptr(stats), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(scale), _ptr(shift),
                                              _ptr(mi), Cn, float(count), float(momentum), float(eps), int(clear), _ptr(num_batches_tracked), _stream()),
              "mhe_bn_finalize_step")
    else:
        check(_lib.lib().mhe_bn_finalize(_ptr(stats), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                         _ptr(scale), _ptr(shift), _ptr(mi), Cn, float(count), float(momentum), float(eps), _stream()),
              "mhe_bn_finalize")
    return (scale, shift, mi) if want_mean_invstd else (scale, shift)


def bn_act(x, scale, shift, res=None, res_scale=None, res_shift=None, relu=True, out=None):
    Cn = x.shape[-1]
    P = x.numel() // Cn
    y = out if out is not None else torch.empty_like(x)
    with _Timed(lambda: "mhe::conv::bn_act_kernel<%s>" % ("float" if x.dtype == torch.float32 else "unsigned short"), 0.0, x.element_size() * x.numel() * (2 + (res is not None))):
        check(_lib.lib().mhe_bn_act_nhwc(_ptr(x), _ptr(scale), _ptr(shift), _ptr(res), _ptr(res_scale), _ptr(res_shift),
                                         _ptr(y), P, Cn, int(relu), dtype_code(x.dtype), _stream()), "mhe_bn_act_nhwc")
    return y


def maxpool3x3s2(x, scale=None, shift=None):
    B, H, W, Cn = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, Cn, device=x.device, dtype=x.dtype)
    check(_lib.lib().mhe_maxpool3x3s2_nhwc(_ptr(x), _ptr(scale), _ptr(shift), _ptr(y), B, H, W, Cn, dtype_code(x.dtype),
                                           _stream()), "mhe_maxpool3x3s2_nhwc")
    return y


def avgpool(x):
    B, H, W, Cn = x.shape
    y = torch.empty(B, Cn, device=x.device, dtype=torch.float32)
    check(_lib.lib().mhe_avgpool_nhwc(_ptr(x), _ptr(y), B, H * W, Cn, dtype_code(x.dtype), _stream()), "mhe_avgpool_nhwc")
    return y


def bn_act_avgpool(x, scale, shift, res=None, rscale=None, rshift=None, relu=True):
    """avgpool(bn_act(x, scale, shift, res, rscale, rshift, relu)) in one pass (mhe_bn_act_avgpool_nhwc): the same bits, no block-wide store"""
    B, H, W, Cn = x.shape
    _chk(x, x.dtype, "bn_act_avgpool.x"); _chk(scale, torch.float32, "bn_act_avgpool.scale", (Cn,)); _chk(shift, torch.float32, "bn_act_avgpool.shift", (Cn,))
    if res is not None:
        _chk(res, x.dtype, "bn_act_avgpool.res", x.shape)
    if rscale is not None:
        _chk(rscale, torch.float32, "bn_act_avgpool.rscale", (Cn,)); _chk(rshift, torch.float32, "bn_act_avgpool.rshift", (Cn,))
    y = torch.empty(B, Cn, device=x.device, dtype=torch.float32)
    check(_lib.lib().mhe_bn_act_avgpool_nhwc(_ptr(x), _ptr(scale), _ptr(shift), _ptr(res), _ptr(rscale), _ptr(rshift), _ptr(y), B, H * W, Cn, int(relu),
                                             dtype_code(x.dtype), _stream()), "mhe_bn_act_avgpool_nhwc")
    return y


def nchw_to_nhwc(x, dtype=torch.float32, cpad=None):
    """NCHW f32 -> NHWC `dtype`, channels zero-padded to a 16-byte chunk (or to cpad: mhe_nchw_to_nhwc_pad)"""
    B, Cn, H, W = x.shape
    _chk(x, torch.float32, "nchw_to_nhwc.x")
    ce = 4 if dtype == torch.float32 else 8
    Cp = (Cn + ce - 1) // ce * ce if cpad is None else int(cpad)
    y = torch.empty(B, H, W, Cp, device=x.device, dtype=dtype)
    if cpad is None:
        check(_lib.lib().mhe_nchw_to_nhwc(_ptr(x), _ptr(y), B, Cn, H, W, dtype_code(dtype), _stream()), "mhe_nchw_to_nhwc")
    else:
        check(_lib.lib().mhe_nchw_to_nhwc_pad(_ptr(x), _ptr(y), B, Cn, Cp, H, W, dtype_code(dtype), _stream()), "mhe_nchw_to_nhwc_pad")
    return y


# ---- train-step (reverse) kernels ---------------------------------------------------------------------
_WGRAD_WS = {}          # per device: one workspace for the partial slabs of the pixel-range split, grown to the largest layer
_WGRAD_WS_RETIRED = []  # superseded workspaces stay alive: a captured HIP graph (train.GraphedStep) may have baked their address in


def _wgrad_ws(device, need):
    ws = _WGRAD_WS.get(device)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _WGRAD_WS_RETIRED.append(ws)
        ws = _WGRAD_WS[device] = torch.empty(need, device=device, dtype=torch.float32)
    return ws
WGRAD_SLABS = True      # False: f32 atomics into dw (the form without a workspace)


_WG_WAVES = {(256, 256): "4, 4" if os.environ.get("MHE_WGRAD_W16", "1") != "0" else "4, 2", (128, 128): "2, 2", (64, 128): "1, 4", (128, 64): "4, 1",
             (64, 64): "2, 2"}


def _wgrad_kernel_name(d, Ho=0, Wo=0, nbatch=1):
    """the weight-gradient instantiation the launcher will pick, spelled as rocprofv3 prints it"""
    v = _lib.lib().mhe_conv_wgrad_variant(C.byref(d), Ho, Wo, nbatch)
    kind, bm, bn = v // 1000000, (v % 1000000) // 1000, v % 1000
    if kind == 1:
        return "mhe::wgrad::wgrad_dma_kernel<%d, %d, %s>" % (bm, bn, _WG_WAVES.get((bm, bn), "?"))
    if kind == 2:
        return "mhe::wgrad::wgrad_bf16_kernel<%d, %d, %s>" % (bm, bn, _WG_WAVES.get((bm, bn), "?"))
    return "mhe::wgrad::wgrad_kernel<%s, %d, %d>" % ("float" if d.dtype == F32 else "unsigned short", bm, bn)


class _Timed:
    """HIP events on the launch stream around one launch, appended to KERNEL_TIMES while TIMING is on (bench.py's live roofline pass)"""
    def __init__(self, name_fn, flops, nbytes):
        self.on = TIMING and name_fn is not None
        if self.on:
            self.name, self.flops, self.nbytes = name_fn(), flops, nbytes
            self.ev0, self.ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.ev0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ev1.record()
            KERNEL_TIMES.append((self.name, self.flops, self.ev0, self.ev1, self.nbytes))
        return False


def _dg_timed(d, x, y, w, residual, mask, mask_bits, bn_ys, tag, xcat=None):
    """tools/train_lines.py: a data-gradient launch under a descriptive name with the bytes it has to move (operand, result, weights,
    residual, the gate - as bits where the kernel reads bits - and the raw tensors of the BatchNorm-reverse sums that are not the gate)"""
    if not (TIMING and TIMING_DG):
        return _Timed(None, 0.0, 0)
    es = x.element_size()
    nb = es * (x.numel() + y.numel() + w.numel() + (residual.numel() if residual is not None else 0) + (xcat.numel() if xcat is not None else 0))
    nb += mask_bits.numel() if mask_bits is not None else es * mask.numel()
    nb += sum(es * b.numel() for b in bn_ys if b is not None and b.data_ptr() != mask.data_ptr())
    name = "dgrad %dx%d s%d %d->%d @%dx%d%s%s bn%d %s" % (d.KH, d.KW, d.stride, d.Cin, d.Cout, y.shape[1], y.shape[2], " +res" if residual is not None else "",
                                                      " half" if d.res_half else "", sum(b is not None for b in bn_ys), tag)
    t = _Timed(lambda: name, 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * d.Cout * d.KH * d.KW * (d.Cin + (xcat.shape[-1] if xcat is not None else 0)), nb)
    return t


def conv_wgrad(x, gy, KH, KW, stride, pad, dw, ldw=0):
    """dw[Cout, KH*KW*Cin] (f32, pre-zeroed or accumulating) += gy^T (*) x ; x [B,H,W,Cin], gy [B,Ho,Wo,Cout]."""
    B, H, W, Cin = x.shape
    Cout = gy.shape[-1]
    dt = x.dtype
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    _chk(x, dt, "wgrad.x"); _chk(gy, dt, "wgrad.gy", (B, Ho, Wo, Cout)); _chk(dw, torch.float32, "wgrad.dw")
    if dw.numel() < (Cout - 1) * (ldw or KH * KW * Cin) + KH * KW * Cin:
        raise ValueError(f"wgrad.dw: {dw.numel()} floats cannot hold [{Cout}, {KH * KW * Cin}] at pitch {ldw or KH * KW * Cin}")
    d = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride, pad, dtype_code(dt), 0, 0)
    L = _lib.lib()
    need = L.mhe_conv_wgrad_workspace_floats(C.byref(d)) if WGRAD_SLABS else 0
    # (timed: the kernel + its slab reducer; algorithmic bytes: both operands once + the gradient)
    with _Timed(lambda: _wgrad_kernel_name(d), 2.0 * B * Ho * Wo * Cout * KH * KW * Cin, x.element_size() * (x.numel() + gy.numel()) + 4 * Cout * KH * KW * Cin):
        if need:
            ws = _wgrad_ws(x.device, need)
            check(L.mhe_conv_wgrad_ws_nhwc(C.byref(d), _ptr(x), _ptr(gy), _ptr(dw), int(ldw), _ptr(ws), ws.numel(), _stream()), "mhe_conv_wgrad_ws_nhwc")
        else:
            check(L.mhe_conv_wgrad_nhwc(C.byref(d), _ptr(x), _ptr(gy), _ptr(dw), int(ldw), _stream()), "mhe_conv_wgrad_nhwc")
    return dw


def conv_wgrad_multi(items):
    """several independent weight gradients in one call (mhe_conv_wgrad_multi_nhwc): items = [(x, gy, KH, KW, stride, pad, dw), ...] as for
    conv_wgrad.  Problems of one tile shape share launches: the chip is filled by the tiles of all of them, each pixel range is cut only as far
    as a common slice length asks (the partial-slab traffic of one launch per layer falls by the number of layers that share the chip).
    Fixed summation order; dW += as in conv_wgrad."""
    if not items:
        return
    if TIMING and len(items) > 1:
        # bench.py's live roofline pass: one call per tile class, so that every timed interval is ONE kernel instantiation (+ its reducer) under
        # the name rocprofv3 prints - the product path makes one call for the whole bucket
        groups = {}
        for it in items:
            x, gy, KH, KW, stride, pad, dw = it
            d = ConvDesc(x.shape[0], x.shape[1], x.shape[2], x.shape[3], gy.shape[-1], KH, KW, stride, pad, dtype_code(x.dtype), 0, 0)
            groups.setdefault(_lib.lib().mhe_conv_wgrad_variant(C.byref(d), 0, 0, 1), []).append(it)
        if len(groups) > 1:
            for g in groups.values():
                conv_wgrad_multi(g)
            return
    n = len(items)
    arr = (_lib.WgradItem * n)()
    flops = nbytes = 0.0
    for i, (x, gy, KH, KW, stride, pad, dw) in enumerate(items):
        B, H, W, Cin = x.shape
        Cout = gy.shape[-1]
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        _chk(x, x.dtype, "wgrad_multi.x"); _chk(gy, x.dtype, "wgrad_multi.gy", (B, Ho, Wo, Cout)); _chk(dw, torch.float32, "wgrad_multi.dw")
        if dw.numel() < Cout * KH * KW * Cin:
            raise ValueError(f"wgrad_multi.dw[{i}]: {dw.numel()} floats cannot hold [{Cout}, {KH * KW * Cin}]")
        arr[i].d = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride, pad, dtype_code(x.dtype), 0, 0)
        arr[i].x, arr[i].gy, arr[i].dw, arr[i].ldw = x.data_ptr(), gy.data_ptr(), dw.data_ptr(), 0
        flops += 2.0 * B * Ho * Wo * Cout * KH * KW * Cin
        nbytes += x.element_size() * (x.numel() + gy.numel()) + 4 * Cout * KH * KW * Cin
    L = _lib.lib()
    need = L.mhe_conv_wgrad_multi_workspace_floats(C.byref(arr), n)
    ws = _wgrad_ws(items[0][0].device, need) if need else None
    def name():
        k = _wgrad_kernel_name(arr[0].d)
        return k.replace("wgrad_dma_kernel", "wgrad_dma_multi_kernel") if n > 1 else k
    with _Timed(name, flops, nbytes):
        check(L.mhe_conv_wgrad_multi_nhwc(C.byref(arr), n, _ptr(ws), ws.numel() if ws is not None else 0, _stream()), "mhe_conv_wgrad_multi_nhwc")


def conv_wgrad_batched(x, gy, dw, dw_batch_stride, nbatch, x_batch_stride=None, gy_batch_stride=None):
    """nbatch dense weight gradients dw_b [N, K] += gy_b [R, N]^T x_b [R, K] in one grouped launch (mhe_conv_wgrad_batched_nhwc).  x / gy: bf16
    tensors whose storage holds the problems at the given element strides (default: x [nbatch, R, K], gy [nbatch, R, N] contiguous); dw: the
    f32 view of the FIRST problem's gradient, the others lie dw_batch_stride floats apart (the train step's raw gradient arena)."""
    R, K = x.shape[-2], x.shape[-1]
    N = gy.shape[-1]
    if x.dtype != torch.bfloat16 or gy.dtype != torch.bfloat16 or not x.is_cuda or not gy.is_cuda or dw.dtype != torch.float32:
        raise _lib.MheError("conv_wgrad_batched: bf16 device operands and an f32 gradient expected")
    xs = R * K if x_batch_stride is None else int(x_batch_stride)
    gs = R * N if gy_batch_stride is None else int(gy_batch_stride)
    d = ConvDesc(R, 1, 1, K, N, 1, 1, 1, 0, BF16, 0, 0)
    L = _lib.lib()
    need = L.mhe_conv_wgrad_batched_workspace_floats(C.byref(d), nbatch) if WGRAD_SLABS else 0
    ws = _wgrad_ws(x.device, need) if need else None
    with _Timed(lambda: _wgrad_kernel_name(d, 0, 0, nbatch), 2.0 * nbatch * R * N * K, nbatch * (2 * R * (K + N) + 4 * N * K)):
        check(L.mhe_conv_wgrad_batched_nhwc(C.byref(d), nbatch, _ptr(x), xs, _ptr(gy), gs, _ptr(dw), int(dw_batch_stride), 0, _ptr(ws),
                                            ws.numel() if ws is not None else 0, _stream()), "mhe_conv_wgrad_batched_nhwc")
    return dw


def conv_wgrad_rect(x, gy, KH, KW, stride_h, stride_w, pad_h, pad_w, dw):
    """dw[Cout, KH*KW*Cin] += gy^T (*) x for a convolution with separate height / width stride and (top / left) padding whose output size is
    gy's (mhe_conv_wgrad_rect_nhwc): the stem's weight gradient over pixel pairs (train.TrainStep)."""
    B, H, W, Cin = x.shape
    Ho, Wo, Cout = gy.shape[1], gy.shape[2], gy.shape[3]
    dt = x.dtype
    _chk(x, dt, "wgrad_rect.x"); _chk(gy, dt, "wgrad_rect.gy", (B, Ho, Wo, Cout)); _chk(dw, torch.float32, "wgrad_rect.dw")
    if dw.numel() < Cout * KH * KW * Cin:
        raise ValueError(f"wgrad_rect.dw: {dw.numel()} floats cannot hold [{Cout}, {KH * KW * Cin}]")
    d = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride_h, pad_h, dtype_code(dt), 0, 0)
    L = _lib.lib()
    need = L.mhe_conv_wgrad_rect_workspace_floats(C.byref(d), Ho, Wo) if WGRAD_SLABS else 0
    ws = None
    if need:
        ws = _wgrad_ws(x.device, need)
    with _Timed(lambda: _wgrad_kernel_name(d, Ho, Wo), 2.0 * B * Ho * Wo * Cout * KH * KW * Cin, x.element_size() * (x.numel() + gy.numel()) + 4 * Cout * KH * KW * Cin):
        check(L.mhe_conv_wgrad_rect_nhwc(C.byref(d), stride_w, pad_w, Ho, Wo, _ptr(x), _ptr(gy), _ptr(dw), 0, _ptr(ws), ws.numel() if ws is not None else 0,
                                         _stream()), "mhe_conv_wgrad_rect_nhwc")
    return dw


def linear_wgrad(x, gy, dw):
    """dw[N,K] += gy[R,N]^T x[R,K] (torch.nn.Linear weight layout), f32."""
    R, K = x.shape
    return conv_wgrad(x.view(R, 1, 1, K), gy.view(R, 1, 1, gy.shape[1]), 1, 1, 1, 0, dw)


_COLSUM_WS = {}
_COLSUM_WS_RETIRED = []     # (a captured HIP graph may have baked a superseded workspace's address in: keep them alive, as _WGRAD_WS_RETIRED)


def colsum(rows, out, group_width=0, group_stride=0):
    """out[c] += sum_r rows[r][c], summed in a fixed order (mhe_colsum_ws_f32); with group_width g: column c is added to
    out[(c // g) * group_stride + c % g] (the l2 bias gradients of the flow's nets lie one raw-gradient slot apart)"""
    Cc = rows.shape[-1]
    R = rows.numel() // Cc
    _chk(rows, rows.dtype, "colsum.rows"); _chk(out, torch.float32, "colsum.out")
    if not group_width and out.numel() < Cc:          # (grouped: `out` is the first group's view into a larger arena)
        raise _lib.MheError(f"colsum.out: {out.numel()} floats cannot hold {Cc} columns")
    L = _lib.lib()
    need = L.mhe_colsum_workspace_floats(R, Cc)
    ws = None
    if need:
        ws = _COLSUM_WS.get(rows.device)
        if ws is None or ws.numel() < need:
            if ws is not None:
                _COLSUM_WS_RETIRED.append(ws)
            ws = _COLSUM_WS[rows.device] = torch.empty(need, device=rows.device, dtype=torch.float32)
    check(L.mhe_colsum_ws_f32(_ptr(rows), _ptr(out), R, Cc, dtype_code(rows.dtype), int(group_width), int(group_stride), _ptr(ws),
                              ws.numel() if ws is not None else 0, _stream()), "mhe_colsum_ws_f32")
    return out


def gather(src, idx, dst, idx2=None):
    """dst[i] = src[idx[i]] (+ src[idx2[i]]), negative index = 0 ; src f32 flat, idx int32, dst f32 or bf16 of idx.numel() elements"""
    _chk(src, torch.float32, "gather.src"); _chk(idx, torch.int32, "gather.idx")
    if idx2 is not None:
        _chk(idx2, torch.int32, "gather.idx2", idx.shape)
    if dst.numel() != idx.numel() or dst.dtype not in (torch.float32, torch.bfloat16) or not dst.is_contiguous():
        raise ValueError("gather.dst: contiguous f32/bf16 tensor with idx.numel() elements expected")
    check(_lib.lib().mhe_gather_f32(_ptr(src), _ptr(idx), _ptr(idx2), _ptr(dst), idx.numel(), dtype_code(dst.dtype), _stream()), "mhe_gather_f32")
    return dst


def affine8(idx, with_bad=False):
    """host side of gather_affine8: int64 index vector (numel a multiple of 8, -1 = zero) -> (base_stride int32 [n/8, 2], mask uint8 [n/8]) if
    every group of eight is src[base + k * stride] on its valid positions, else None.  with_bad: always the pair plus a bool vector of the
    groups that are NOT affine (their entries are meaningless: the caller gathers those groups by index)"""
    I = idx.reshape(-1, 8).to(torch.int64)
    valid = I >= 0
    k = torch.arange(8, dtype=torch.int64)
    nv = valid.sum(1)
    first = torch.where(valid, k, torch.full_like(k, 8)).min(1).values.clamp(max=7)                 # first valid position
    second = torch.where(valid & (k > first[:, None]), k, torch.full_like(k, 8)).min(1).values      # next valid position (8: none)
    rows = torch.arange(I.shape[0])
    v0 = I[rows, first]
    has2 = second < 8
    v1 = I[rows, second.clamp(max=7)]
    num = v1 - v0
    den = (second - first).clamp(min=1)
    stride = torch.where(has2, torch.div(num, den, rounding_mode="floor"), torch.zeros_like(num))
    base = torch.where(nv > 0, v0 - first * stride, torch.zeros_like(v0))
    pred = base[:, None] + k[None, :] * stride[:, None]
    bad = ~(((pred == I) | ~valid).all(1)) | (stride.abs() >= 2 ** 31) | (base < -2 ** 31) | (base >= 2 ** 31)
    if not with_bad and bool(bad.any()):
        return None
    base, stride = torch.where(bad, torch.zeros_like(base), base), torch.where(bad, torch.zeros_like(stride), stride)
    mask = torch.where(bad, torch.zeros_like(nv), (valid.to(torch.int64) << k[None, :]).sum(1)).to(torch.uint8)
    enc = torch.stack([base, stride], 1).to(torch.int32).contiguous(), mask.contiguous()
    return enc + (bad,) if with_bad else enc


def gather_affine8(src, base_stride, mask, dst):
    """dst[8 g + k] = bf16(src[base_g + k stride_g]) where bit k of mask[g] is set, else 0 (mhe_gather_affine8_bf16)"""
    n8 = mask.numel()
    _chk(src, torch.float32, "gather.src"); _chk(base_stride, torch.int32, "gather.base_stride", (n8, 2)); _chk(mask, torch.uint8, "gather.mask", (n8,))
    if dst.numel() != 8 * n8 or dst.dtype != torch.bfloat16 or not dst.is_contiguous():
        raise ValueError("gather_affine8.dst: contiguous bf16 tensor of 8 x mask.numel() elements expected")
    check(_lib.lib().mhe_gather_affine8_bf16(_ptr(src), _ptr(base_stride), _ptr(mask), _ptr(dst), 8 * n8, _stream()), "mhe_gather_affine8_bf16")
    return dst


def flow_mask_pad(x, mask_row, out):
    R, dim = x.shape
    _chk(x, torch.float32, "mask_pad.x"); _chk(mask_row, torch.float32, "mask_pad.mask", (dim,)); _chk(out, torch.float32, "mask_pad.out", (R, 64))
    check(_lib.lib().mhe_flow_mask_pad_f32(_ptr(x), _ptr(mask_row), _ptr(out), R, dim, _stream()), "mhe_flow_mask_pad_f32")
    return out


def flow_mask_pad_mixed(x, mask_row, out_f32=None, out_bf16=None):
    R, dim = x.shape
    _chk(x, torch.float32, "mask_pad.x"); _chk(mask_row, torch.float32, "mask_pad.mask", (dim,))
    if out_f32 is not None:
        _chk(out_f32, torch.float32, "mask_pad.out", (R, 64))
    if out_bf16 is not None:
        _chk(out_bf16, torch.bfloat16, "mask_pad.out_bf16", (R, 64))
    check(_lib.lib().mhe_flow_mask_pad_mixed(_ptr(x), _ptr(mask_row), _ptr(out_f32), _ptr(out_bf16), R, dim, _stream()), "mhe_flow_mask_pad_mixed")


def flow_lrelu_bwd_sum(g, h, N, B, sum_out, sum_stride, out_f32=None, out_bf16=None, slope=0.01, sum_out_t=None):
    """out = g * (h > 0 ? 1 : slope) (f32 and / or bf16) and sum_out[b] = sum over the image's N hypothesis rows; sum_out may be a
    column slice of a wider [B, sum_stride] matrix"""
    R, H = g.shape
    _chk(g, g.dtype, "lrelu_bwd_sum.g", (N * B, H)); _chk(h, h.dtype, "lrelu_bwd_sum.h", (R, H))
    if out_f32 is not None:
        _chk(out_f32, torch.float32, "lrelu_bwd_sum.out_f32", (R, H))
    if out_bf16 is not None:
        _chk(out_bf16, torch.bfloat16, "lrelu_bwd_sum.out_bf16", (R, H))
    check(_lib.lib().mhe_flow_lrelu_bwd_sum(_ptr(g), dtype_code(g.dtype), _ptr(h), dtype_code(h.dtype), _ptr(out_f32), _ptr(out_bf16),
                                            C.c_void_p(sum_out.data_ptr()), int(sum_stride), C.c_void_p(0 if sum_out_t is None else sum_out_t.data_ptr()),
                                            N, B, H, float(slope), _stream()), "mhe_flow_lrelu_bwd_sum")


def flow_cond_lrelu(P, cond_slice, cond_stride, B):
    """P[r] = leaky_relu(P[r] + cond_slice[(r % B) * cond_stride : +H]) in place; cond_slice = view starting at the net/layer's column"""
    R, H = P.shape
    _chk(P, torch.float32, "cond_lrelu.P")
    check(_lib.lib().mhe_flow_cond_lrelu_f32(_ptr(P), cond_slice.data_ptr(), int(cond_stride), R, B, H, _stream()), "mhe_flow_cond_lrelu_f32")
    return P


def flow_lrelu_bwd(G, Hact, slope=0.01):
    _chk(G, torch.float32, "lrelu_bwd.G"); _chk(Hact, torch.float32, "lrelu_bwd.H", G.shape)
    check(_lib.lib().mhe_flow_lrelu_bwd_f32(_ptr(G), _ptr(Hact), G.numel(), float(slope), _stream()), "mhe_flow_lrelu_bwd_f32")
    return G


def add(a, b, out=None):
    _chk(a, torch.float32, "add.a"); _chk(b, torch.float32, "add.b", a.shape)
    out = a if out is None else out
    check(_lib.lib().mhe_add_f32(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "mhe_add_f32")
    return out


def flow_couple_bwd(x_out, Os, Ot, mask_row, g_out, g_log_p, q_weight, B, x_in, GOs, GOt, g_part, GOs_bf16=None, GOt_bf16=None, db_s=None,
                    db_t=None):
    """db_s / db_t (optional, [64] f32): += the column sums of GOs / GOt (the l2 bias gradients of the s and t nets)"""
    R, dim = x_out.shape
    if GOs_bf16 is not None:
        _chk(GOs_bf16, torch.bfloat16, "couple_bwd.GOs_bf16", (R, 64)); _chk(GOt_bf16, torch.bfloat16, "couple_bwd.GOt_bf16", (R, 64))
    for t, n, s in ((x_out, "x_out", (R, dim)), (Os, "Os", (R, 64)), (Ot, "Ot", (R, 64)), (g_out, "g_out", (R, dim)),
                    (x_in, "x_in", (R, dim)), (GOs, "GOs", (R, 64)), (GOt, "GOt", (R, 64)), (g_part, "g_part", (R, dim))):
        _chk(t, torch.float32, "couple_bwd." + n, s)
    # the kernel's own column sums are f32 atomics from every 64-row workgroup: order-dependent beyond one workgroup - there the sums are
    # taken from the written GOs / GOt rows in a fixed order instead (two more launches on a fallback path)
    in_kernel = db_s is not None and R <= 64
    check(_lib.lib().mhe_flow_couple_bwd_mixed(_ptr(x_out), _ptr(Os), _ptr(Ot), _ptr(mask_row), _ptr(g_out), _ptr(g_log_p),
                                               float(q_weight), _ptr(x_in), _ptr(GOs), _ptr(GOt), _ptr(g_part), _ptr(GOs_bf16), _ptr(GOt_bf16),
                                               _ptr(db_s if in_kernel else None), _ptr(db_t if in_kernel else None), R, B, dim, _stream()),
          "mhe_flow_couple_bwd_mixed")
    if db_s is not None and not in_kernel:
        colsum(GOs, db_s); colsum(GOt, db_t)


def mfma_fragment_major(t):
    """[rows][K] (rows % 16 == 0, K % 32 == 0) -> the same elements in the order mhe_flow_reverse_chain_bf16 reads its operands:
    [rows / 16][K / 32][k-group 4][row 16][8] (lane = k-group * 16 + row of v_mfma_f32_16x16x32_bf16); works on index tensors too"""
    rows, K = t.shape
    return t.reshape(rows // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


def flow_reverse_chain_supported(R, B, dim, hidden, ncoup):
    return bool(_lib.lib().mhe_flow_reverse_chain_supported(R, B, dim, hidden, ncoup))


def flow_sign_bits(h1, h2, B):
    """the signs of the kept activations h1, h2 (bf16 [nets, 64 B, 512], rows n B + b) in the layout mhe_flow_reverse_chain_bf16 reads and
    mhe_flow_couplings_frag_bf16 writes: int32 [nets, B, 2 layers, 8 waves, 64 lanes, 2] - lane (q, l15) of wave w holds bit
    (nt * 4 + mt) * 4 + e = [h[row mt * 16 + l15 of image b][unit 64 w + 16 nt + 4 q + e] > 0] (low word: nt < 2)"""
    nets = h1.shape[0]
    out = []
    for h in (h1, h2):
        t = (h.float() > 0).view(nets, 4, 16, B, 8, 4, 4, 4)                  # [net, mt, l15, b, w, nt, q, e]
        t = t.permute(0, 3, 4, 6, 2, 5, 1, 7).reshape(nets, B, 8, 64, 2, 32).to(torch.int64)      # [net, b, w, lane = q * 16 + l15, half, bit (nt % 2, mt, e)]
        wts = (1 << torch.arange(32, device=h.device, dtype=torch.int64))
        words = (t * wts).sum(-1)                                              # < 2^32
        out.append(torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32))
    return torch.stack(out, 2).contiguous()                                    # [net, b, layer, w, lane, 2]


_DB2_ROWS = {}


def flow_reverse_chain(x_out, g_x, g_logp, q_weight, mask, o_pre, sign_bits, w2F, w1F, w0F, w_net_stride, GOb, G2b, G1b, XPb, Gc, db2,
                       db_net_stride, z0):
    """the RealNVP reverse pass's data-gradient chain over all couplings in one launch (mhe_flow_reverse_chain_bf16; csrc/flow_rev.hip);
    sign_bits: flow_couplings_frag(..., sign_bits=) / flow_sign_bits"""
    R, dim = x_out.shape
    B = Gc.shape[0]
    nets, hidden = o_pre.shape[0], 512
    _chk(x_out, torch.float32, "rev_chain.x_out"); _chk(g_x, torch.float32, "rev_chain.g_x", (R, dim))
    _chk(mask, torch.float32, "rev_chain.mask", (nets // 2, dim)); _chk(o_pre, torch.float32, "rev_chain.o", (nets, R, 64))
    _chk(sign_bits, torch.int32, "rev_chain.sign_bits", (nets, B, 2, 8, 64, 2))
    _chk(GOb, torch.bfloat16, "rev_chain.GO", (nets, R, 64)); _chk(G2b, torch.bfloat16, "rev_chain.G2", (nets, R, hidden))
    _chk(G1b, torch.bfloat16, "rev_chain.G1", (nets, R, hidden)); _chk(XPb, torch.bfloat16, "rev_chain.XP", (nets // 2, R, 64))
    _chk(Gc, torch.float32, "rev_chain.Gc"); _chk(z0, torch.float32, "rev_chain.z0", (R, dim))
    if g_logp is not None:
        _chk(g_logp, torch.float32, "rev_chain.g_logp", (B,))
    rows = _DB2_ROWS.get((x_out.device, B, nets))
    if rows is None:
        rows = _DB2_ROWS[(x_out.device, B, nets)] = torch.empty(B, nets * 64, device=x_out.device, dtype=torch.float32)
    check(_lib.lib().mhe_flow_reverse_chain_bf16(_ptr(x_out), _ptr(g_x), _ptr(g_logp), float(q_weight), _ptr(mask), _ptr(o_pre), _ptr(sign_bits), _ptr(w2F),
                                                 _ptr(w1F), _ptr(w0F), int(w_net_stride), _ptr(GOb), _ptr(G2b), _ptr(G1b), _ptr(XPb), _ptr(Gc),
                                                 Gc.shape[1], _ptr(rows), _ptr(z0), R, B, dim, hidden, nets // 2, _stream()),
          "mhe_flow_reverse_chain_bf16")
    # db2 (+ net * db_net_stride) += the sum over images of the kernel's per-image rows, in a fixed order
    colsum(rows, db2, group_width=64, group_stride=db_net_stride)


def pack_transpose_bf16(src, out=None, outT=None, want_rows=True):
    """src [R][C] f32 -> (bf16 copy [R][C] | None, bf16 transpose [C][R]) in one launch (mhe_pack_transpose_bf16)"""
    R, Cc = src.shape
    _chk(src, torch.float32, "pack_transpose.src")
    if want_rows and out is None:
        out = torch.empty(R, Cc, device=src.device, dtype=torch.bfloat16)
    if outT is None:
        outT = torch.empty(Cc, R, device=src.device, dtype=torch.bfloat16)
    if out is not None:
        _chk(out, torch.bfloat16, "pack_transpose.out", (R, Cc))
    _chk(outT, torch.bfloat16, "pack_transpose.outT", (Cc, R))
    check(_lib.lib().mhe_pack_transpose_bf16(_ptr(src), src.stride(0), _ptr(out), _ptr(outT), R, Cc, _stream()), "mhe_pack_transpose_bf16")
    return out, outT


def flow_couple_accum(g_part, GXs, GXt, mask_row, g_in):
    R, dim = g_part.shape
    check(_lib.lib().mhe_flow_couple_accum_f32(_ptr(g_part), _ptr(GXs), _ptr(GXt), _ptr(mask_row), _ptr(g_in), R, dim, _stream()),
          "mhe_flow_couple_accum_f32")
    return g_in


def bn_mean_invstd(stats, count, eps=1e-5):
    Cc = stats.shape[-1]
    _chk_stats(stats, "bn_mean_invstd.stats", Cc)
    mi = torch.empty(2, Cc, device=stats.device, dtype=torch.float32)
    check(_lib.lib().mhe_bn_mean_invstd(_ptr(stats), _ptr(mi), Cc, float(count), float(eps), _stream()), "mhe_bn_mean_invstd")
    return mi


def bn_backward(g, a, y, mean_invstd, gamma, stats, dgamma, dbeta, want_masked=False, out=None, reduced=False, coef_only=False):
    """train-mode BatchNorm(+ReLU) reverse: returns gy (and g [a>0] when want_masked); writes dgamma / dbeta.
    reduced=True: `stats` already holds the sums (accumulated by the epilogue of the kernel that produced g)."""
    Cc = y.shape[-1]
    P = y.numel() // Cc
    dt = y.dtype
    _chk(g, dt, "bn_bwd.g", y.shape); _chk(y, dt, "bn_bwd.y")
    if a is not None:
        _chk(a, dt, "bn_bwd.a", y.shape)
    _chk_stats(stats, "bn_bwd.stats", Cc); _chk(mean_invstd, torch.float32, "bn_bwd.mean_invstd", (2, Cc))
    _chk(gamma, torch.float32, "bn_bwd.gamma", (Cc,)); _chk(dgamma, torch.float32, "bn_bwd.dgamma", (Cc,)); _chk(dbeta, torch.float32, "bn_bwd.dbeta", (Cc,))
    L = _lib.lib()
    if not reduced:
        check(L.mhe_bn_bwd_reduce_nhwc(_ptr(g), _ptr(a), _ptr(y), _ptr(mean_invstd), _ptr(stats), P, Cc, dtype_code(dt), _stream()), "mhe_bn_bwd_reduce_nhwc")
    coef = torch.empty(3, Cc, device=y.device, dtype=torch.float32)
    check(L.mhe_bn_bwd_finalize(_ptr(stats), _ptr(gamma), _ptr(mean_invstd), _ptr(dgamma), _ptr(dbeta), _ptr(coef), Cc, float(P), _stream()), "mhe_bn_bwd_finalize")
    if coef_only:              # the consumer applies gy = k2 g + k1 y + k0 itself (conv1x1_dgrad_bn_apply)
        return coef
    gy = out if out is not None else torch.empty_like(y)
    gm = torch.empty_like(y) if want_masked else None
    with _Timed(lambda: "mhe::tb::bn_bwd_apply_wide_kernel<%s, %d>" % ("float" if dt == torch.float32 else "unsigned short", 4 if y.numel() * y.element_size() <= (80 << 20) else 1), 0.0, y.element_size() * y.numel() * (3 + (a is not None) + (gm is not None))):
        check(L.mhe_bn_bwd_apply_nhwc(_ptr(g), _ptr(a), _ptr(y), _ptr(coef), _ptr(gy), _ptr(gm), P, Cc, dtype_code(dt), _stream()), "mhe_bn_bwd_apply_nhwc")
    return (gy, gm) if want_masked else gy


def bn_bwd_coef(stats, gamma, mean_invstd, dgamma, dbeta, count):
    """finish a BatchNorm reverse whose sums are in `stats`: writes dgamma / dbeta, returns coef = k2 | k1 | k0 [3, C] of gy = k2 g + k1 y + k0"""
    Cc = gamma.shape[0]
    _chk_stats(stats, "bn_bwd.stats", Cc); _chk(mean_invstd, torch.float32, "bn_bwd.mean_invstd", (2, Cc))
    _chk(gamma, torch.float32, "bn_bwd.gamma", (Cc,)); _chk(dgamma, torch.float32, "bn_bwd.dgamma", (Cc,)); _chk(dbeta, torch.float32, "bn_bwd.dbeta", (Cc,))
    coef = torch.empty(3, Cc, device=stats.device, dtype=torch.float32)
    check(_lib.lib().mhe_bn_bwd_finalize(_ptr(stats), _ptr(gamma), _ptr(mean_invstd), _ptr(dgamma), _ptr(dbeta), _ptr(coef), Cc, float(count), _stream()),
          "mhe_bn_bwd_finalize")
    return coef


def maxpool3x3s2_idx_win(x, scale, shift):
    """maxpool3x3s2_idx(x, scale, shift) that also returns the RAW input at every winner (mhe_maxpool3x3s2_idx_affine_win_nhwc): what
    pooled_bn_sums needs of the full-resolution tensor"""
    B, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, Cc, device=x.device, dtype=x.dtype)
    xwin = torch.empty_like(y)
    idx = torch.empty(B, Ho, Wo, Cc, device=x.device, dtype=torch.uint8)
    _chk(x, x.dtype, "maxpool_idx.x"); _chk(scale, torch.float32, "maxpool_idx.scale", (Cc,)); _chk(shift, torch.float32, "maxpool_idx.shift", (Cc,))
    check(_lib.lib().mhe_maxpool3x3s2_idx_affine_win_nhwc(_ptr(x), _ptr(scale), _ptr(shift), _ptr(y), _ptr(idx), _ptr(xwin), B, H, W, Cc, dtype_code(x.dtype),
                                                          _stream()), "mhe_maxpool3x3s2_idx_affine_win_nhwc")
    return y, idx, xwin


def pooled_bn_sums(g, pooled, xwin, mean_invstd, stats):
    """the stem's BatchNorm-reverse sums from the pooled tensors alone (mhe_pooled_bn_sums_nhwc): g = the gradient of the pooled output,
    pooled = maxpool(relu(bn(y))) (its gate), xwin = the raw y at the winners; adds sum g', sum g' xhat per channel to `stats`"""
    Cc = g.shape[-1]
    dt = g.dtype
    _chk(g, dt, "pooled_bn_sums.g"); _chk(pooled, dt, "pooled_bn_sums.pooled", g.shape); _chk(xwin, dt, "pooled_bn_sums.xwin", g.shape)
    _chk(mean_invstd, torch.float32, "pooled_bn_sums.mean_invstd", (2, Cc)); _chk_stats(stats, "pooled_bn_sums.stats", Cc)
    check(_lib.lib().mhe_pooled_bn_sums_nhwc(_ptr(g), _ptr(pooled), _ptr(xwin), _ptr(mean_invstd), _ptr(stats), g.numel() // Cc, Cc, dtype_code(dt), _stream()),
          "mhe_pooled_bn_sums_nhwc")
    return stats


def maxpool3x3s2_idx(x, scale=None, shift=None):
    """(pooled, winning taps) of a 3x3 / stride-2 / pad-1 max pool; with scale / shift: of relu(x * scale + shift), evaluated on the load
    (mhe_maxpool3x3s2_idx_affine_nhwc: the normalised activation is never materialised)"""
    B, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, Cc, device=x.device, dtype=x.dtype)
    idx = torch.empty(B, Ho, Wo, Cc, device=x.device, dtype=torch.uint8)
    _chk(x, x.dtype, "maxpool_idx.x")
    if scale is not None:
        _chk(scale, torch.float32, "maxpool_idx.scale", (Cc,)); _chk(shift, torch.float32, "maxpool_idx.shift", (Cc,))
        check(_lib.lib().mhe_maxpool3x3s2_idx_affine_nhwc(_ptr(x), _ptr(scale), _ptr(shift), _ptr(y), _ptr(idx), B, H, W, Cc, dtype_code(x.dtype), _stream()),
              "mhe_maxpool3x3s2_idx_affine_nhwc")
        return y, idx
    check(_lib.lib().mhe_maxpool3x3s2_idx_nhwc(_ptr(x), _ptr(y), _ptr(idx), B, H, W, Cc, dtype_code(x.dtype), _stream()), "mhe_maxpool3x3s2_idx_nhwc")
    return y, idx


def maxpool3x3s2_bwd_bn(gy, idx, y, scale, shift, mean_invstd, stats, want_gx=True):
    """reverse of maxpool(relu(bn(y))) up to the BatchNorm's sums: returns gx = scatter(gy, idx) [relu(y * scale + shift) > 0] and adds
    sum gx, sum gx * xhat per channel to `stats` (then bn_backward(gx, None, y, ..., stats, reduced=True) finishes the BatchNorm reverse)"""
    B, Ho, Wo, Cc = gy.shape
    dt = gy.dtype
    _chk(gy, dt, "maxpool_bwd_bn.gy"); _chk(idx, torch.uint8, "maxpool_bwd_bn.idx", gy.shape)
    _chk(y, dt, "maxpool_bwd_bn.y")
    H, W = y.shape[1], y.shape[2]
    if y.shape[0] != B or y.shape[3] != Cc or ((H - 1) // 2 + 1, (W - 1) // 2 + 1) != (Ho, Wo):
        raise ValueError(f"maxpool_bwd_bn: y {tuple(y.shape)} does not pool to gy {tuple(gy.shape)}")
    _chk(scale, torch.float32, "maxpool_bwd_bn.scale", (Cc,)); _chk(shift, torch.float32, "maxpool_bwd_bn.shift", (Cc,))
    _chk(mean_invstd, torch.float32, "maxpool_bwd_bn.mean_invstd", (2, Cc)); _chk_stats(stats, "maxpool_bwd_bn.stats", Cc)
    gx = torch.empty_like(y) if want_gx else None
    check(_lib.lib().mhe_maxpool3x3s2_bwd_bn_nhwc(_ptr(gy), _ptr(idx), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean_invstd), _ptr(stats), _ptr(gx),
                                                  B, H, W, Cc, dtype_code(dt), _stream()), "mhe_maxpool3x3s2_bwd_bn_nhwc")
    return gx


def maxpool3x3s2_bwd_bn_apply(gy, idx, y, scale, shift, mean_invstd, coef):
    """the BatchNorm reverse's result k2 gx + k1 y + k0 for gx = scatter(gy, idx) [relu(y * scale + shift) > 0], gx itself never stored
    (mhe_maxpool3x3s2_bwd_bn_apply_nhwc; coef from bn_backward(..., coef_only=True) on the sums of maxpool3x3s2_bwd_bn(want_gx=False))"""
    B, Ho, Wo, Cc = gy.shape
    dt = gy.dtype
    _chk(gy, dt, "maxpool_bwd_apply.gy"); _chk(idx, torch.uint8, "maxpool_bwd_apply.idx", gy.shape); _chk(y, dt, "maxpool_bwd_apply.y")
    H, W = y.shape[1], y.shape[2]
    _chk(scale, torch.float32, "maxpool_bwd_apply.scale", (Cc,)); _chk(shift, torch.float32, "maxpool_bwd_apply.shift", (Cc,))
    _chk(mean_invstd, torch.float32, "maxpool_bwd_apply.mean_invstd", (2, Cc)); _chk(coef, torch.float32, "maxpool_bwd_apply.coef", (3, Cc))
    out = torch.empty_like(y)
    check(_lib.lib().mhe_maxpool3x3s2_bwd_bn_apply_nhwc(_ptr(gy), _ptr(idx), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean_invstd), _ptr(coef), _ptr(out),
                                                        B, H, W, Cc, dtype_code(dt), _stream()), "mhe_maxpool3x3s2_bwd_bn_apply_nhwc")
    return out


def maxpool3x3s2_bwd(gy, idx, H, W):
    B, Ho, Wo, Cc = gy.shape
    _chk(gy, gy.dtype, "maxpool_bwd.gy"); _chk(idx, torch.uint8, "maxpool_bwd.idx", gy.shape)
    gx = torch.empty(B, H, W, Cc, device=gy.device, dtype=gy.dtype)
    check(_lib.lib().mhe_maxpool3x3s2_bwd_nhwc(_ptr(gy), _ptr(idx), _ptr(gx), B, H, W, Cc, dtype_code(gy.dtype), _stream()), "mhe_maxpool3x3s2_bwd_nhwc")
    return gx


def avgpool_bwd(g, HW, dtype, mask=None):
    """gx[b,p,c] = g[b,c] / HW, zeroed where mask[b,p,c] <= 0 (ReLU gate of the pooled tensor) when a mask is given"""
    B, Cc = g.shape
    _chk(g, torch.float32, "avgpool_bwd.g")
    gx = torch.empty(B, HW, Cc, device=g.device, dtype=dtype)
    if mask is not None:
        _chk(mask, dtype, "avgpool_bwd.mask")
        if mask.numel() != gx.numel():
            raise ValueError("avgpool_bwd.mask: wrong size")
    check(_lib.lib().mhe_avgpool_bwd_nhwc(_ptr(g), _ptr(mask), _ptr(gx), B, HW, Cc, dtype_code(dtype), _stream()), "mhe_avgpool_bwd_nhwc")
    return gx


def upsample2(g, H, W, base=None):
    """out[b,2i,2j] = g[b,i,j] (+ base), zero (+ base) elsewhere; out [B,H,W,C]"""
    B, Ho, Wo, Cc = g.shape
    if (Ho, Wo) != ((H + 1) // 2, (W + 1) // 2):
        raise ValueError(f"upsample2: g {tuple(g.shape)} does not match output {H}x{W}")
    _chk(g, g.dtype, "upsample2.g")
    if base is not None:
        _chk(base, g.dtype, "upsample2.base", (B, H, W, Cc))
    out = torch.empty(B, H, W, Cc, device=g.device, dtype=g.dtype)
    check(_lib.lib().mhe_upsample2_nhwc(_ptr(g), _ptr(base), _ptr(out), B, H, W, Cc, dtype_code(g.dtype), _stream()), "mhe_upsample2_nhwc")
    return out


_SQ_WS = {}


def sqnorm(g, out):
    """out[0] = |g|^2, summed in a fixed order (bitwise reproducible for equal inputs)"""
    _chk(g, torch.float32, "sqnorm.g"); _chk(out, torch.float32, "sqnorm.out", (1,))
    ws = _SQ_WS.get(g.device)
    if ws is None:
        ws = _SQ_WS[g.device] = torch.empty(_lib.lib().mhe_sqnorm_workspace_floats(), device=g.device, dtype=torch.float32)
    check(_lib.lib().mhe_sqnorm_f32(_ptr(g), g.numel(), _ptr(ws), _ptr(out), _stream()), "mhe_sqnorm_f32")
    return out


def train_tick(step, sq):
    _chk(step, torch.int32, "tick.step", (1,)); _chk(sq, torch.float32, "tick.sqnorm", (1,))
    check(_lib.lib().mhe_train_tick(_ptr(step), _ptr(sq), _stream()), "mhe_train_tick")


def adam_step(p, g, m, v, sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, max_norm=1.0, grad_scale=1.0):
    n = p.numel()
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, torch.float32, "adam." + nm, (n,))
    check(_lib.lib().mhe_adam_step_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, _ptr(sq), _ptr(step), float(lr), float(beta1), float(beta2),
                                       float(eps), float(max_norm), float(grad_scale), _stream()), "mhe_adam_step_f32")


def flow_cond_lrelu_mixed(pre, cond_slice, cond_stride, B, out_f32=None, out_bf16=None):
    """leaky_relu(pre + cond[r % B]) from f32 or bf16 `pre` into f32 and / or bf16 outputs"""
    R, H = pre.shape
    _chk(pre, pre.dtype, "cond_lrelu_mixed.pre")
    if out_f32 is not None:
        _chk(out_f32, torch.float32, "cond_lrelu_mixed.out_f32", (R, H))
    if out_bf16 is not None:
        _chk(out_bf16, torch.bfloat16, "cond_lrelu_mixed.out_bf16", (R, H))
    check(_lib.lib().mhe_flow_cond_lrelu_mixed(_ptr(pre), dtype_code(pre.dtype), cond_slice.data_ptr(), int(cond_stride), _ptr(out_f32),
                                               _ptr(out_bf16), R, B, H, _stream()), "mhe_flow_cond_lrelu_mixed")


def flow_lrelu_bwd_mixed(g, h, out_f32=None, out_bf16=None, slope=0.01):
    _chk(g, g.dtype, "lrelu_bwd_mixed.g"); _chk(h, h.dtype, "lrelu_bwd_mixed.h", g.shape)
    check(_lib.lib().mhe_flow_lrelu_bwd_mixed(_ptr(g), dtype_code(g.dtype), _ptr(h), dtype_code(h.dtype), _ptr(out_f32), _ptr(out_bf16),
                                              g.numel(), float(slope), _stream()), "mhe_flow_lrelu_bwd_mixed")


# ---- conditional Glow: the ActNorm + LU re-parameterisation on the device (csrc/glow_affine.hip) ---------------------------------------
def glow_affine(ptr_table, layers, features, eps, out=None):
    """ptr_table: int64 [layers, 6] device tensor of parameter addresses (log_scale, shift, lower, upper, unconstrained diag, bias).
    Returns / refreshes dict(A, Ainv, AinvT [L,64,64], c, cinv [L,64], const_parts [L], ws float64): mhe_glow_affine_f64"""
    dev = ptr_table.device
    if out is None:
        out = {k: torch.zeros(layers, 64, 64, device=dev) for k in ("A", "Ainv", "AinvT")}
        out.update({k: torch.zeros(layers, 64, device=dev) for k in ("c", "cinv")})
        out["const_parts"] = torch.zeros(layers, device=dev)
        out["ws"] = torch.zeros(int(_lib.lib().mhe_glow_affine_workspace_doubles(layers, features)), device=dev, dtype=torch.float64)
    check(_lib.lib().mhe_glow_affine_f64(_ptr(ptr_table), layers, features, float(eps), _ptr(out["A"]), _ptr(out["c"]), _ptr(out["Ainv"]),
                                         _ptr(out["AinvT"]), _ptr(out["cinv"]), _ptr(out["const_parts"]), _ptr(out["ws"]), _stream()), "mhe_glow_affine_f64")
    return out


def glow_reparam_bwd(g_ainv_ptrs, g_cinv_ptrs, g_logp, layers, features, ws, grad_ptrs, q_sign=-1.0):
    """gradients of every layer's six small tensors, written through grad_ptrs (int64 [layers, 6] addresses): mhe_glow_reparam_bwd_f64"""
    check(_lib.lib().mhe_glow_reparam_bwd_f64(_ptr(g_ainv_ptrs), _ptr(g_cinv_ptrs), _ptr(g_logp), 0 if g_logp is None else g_logp.numel(), float(q_sign),
                                              layers, features, _ptr(ws), _ptr(grad_ptrs), _stream()), "mhe_glow_reparam_bwd_f64")


def glow_finish(z_padded, v_padded, logdet, R, dim, inverse, const_parts, want_out=True):
    """(out [R,dim] | None, log_prob [R]) with the log-determinant constant summed from the device-resident per-layer parts"""
    out = torch.empty(R, dim, device=z_padded.device) if want_out else None
    logp = torch.empty(R, device=z_padded.device)
    check(_lib.lib().mhe_glow_finish_dev_f32(_ptr(z_padded), _ptr(v_padded), _ptr(logdet), _ptr(out), _ptr(logp), R, dim, -1.0 if inverse else 1.0,
                                             _ptr(const_parts), const_parts.numel(), _stream()), "mhe_glow_finish_dev_f32")
    return out, logp


# ---- conditional Glow: the sampling direction of all layers in one launch (csrc/glow_fwd.hip) -------------------------------------------
def dropout_bits(n, p, device, state=None):
    """keep bits of n elements (uint8 [n / 8], the format dropout_ writes / applies), drawn on the device: mhe_dropout_bits"""
    bits = torch.empty(n // 8, device=device, dtype=torch.uint8)
    st = state if state is not None else rng_state(device)
    check(_lib.lib().mhe_dropout_bits(_ptr(bits), n, float(p), _ptr(st), _stream()), "mhe_dropout_bits")
    return bits


def glow_layers_supported(N, B, dim, hidden, layers, blocks):
    return bool(_lib.lib().mhe_glow_layers_supported(N, B, dim, hidden, layers, blocks))


def glow_fused_layout(wx, w0, w1, wf, bf, b0, b1, dim):
    """the weight operands of mhe_glow_layers_bf16 from per-layer stacks - wx [L, 512, 64] (initial layer on the padded variable), w0 / w1
    [L, 2, 512, 512], wf [L, >= 2 T, 512] / bf [L, >= 2 T] (final layer, rows [shift (T) | unconstrained scale (T)]), b0 / b1 [L, 2, 512] - as
    VALUES (float tensors: zero fill) or as INDEX tensors into a flat parameter buffer (integer tensors: -1 fill; the trainer's gather tables).
    The final layer's rows move to the flow variable's own columns: row c of wsF / wuF = the shift / scale row of transform column c."""
    L = wx.shape[0]
    idx = not wx.is_floating_point()
    fill = -1 if idx else 0
    ws, wu = (torch.full((L, 64, 512), fill, dtype=wf.dtype, device=wf.device) for _ in range(2))
    bs, bu = (torch.full((L, 64), fill, dtype=bf.dtype, device=bf.device) for _ in range(2))
    for l in range(L):
        first = 1 - (l & 1)
        T = dim // 2 if first else (dim + 1) // 2
        cols = torch.arange(first, dim, 2, device=wf.device)
        ws[l, cols], wu[l, cols] = wf[l, :T], wf[l, T:2 * T]
        bs[l, cols], bu[l, cols] = bf[l, :T], bf[l, T:2 * T]
    fr = mfma_fragment_major
    tr = lambda m: m.t().contiguous()
    rev = {"wsT": torch.stack([fr(tr(ws[l])) for l in range(L)]), "wuT": torch.stack([fr(tr(wu[l])) for l in range(L)]),
           "w0T": torch.stack([torch.stack([fr(tr(w0[l, b])) for b in range(2)]) for l in range(L)]),
           "w1T": torch.stack([torch.stack([fr(tr(w1[l, b])) for b in range(2)]) for l in range(L)]),
           "wxT": torch.stack([fr(tr(wx[l])) for l in range(L)])}                       # operands of the reverse chain (mhe_glow_reverse_chain_bf16)
    return {**rev, "wxF": torch.stack([fr(wx[l]) for l in range(L)]), "w0F": torch.stack([torch.stack([fr(w0[l, b]) for b in range(2)]) for l in range(L)]),
            "w1F": torch.stack([torch.stack([fr(w1[l, b]) for b in range(2)]) for l in range(L)]),
            "wsF": torch.stack([fr(ws[l]) for l in range(L)]), "wuF": torch.stack([fr(wu[l]) for l in range(L)]),
            "bs": bs, "bu": bu, "b0": b0.contiguous(), "b1": b1.contiguous()}


def glow_layers(noise, ctab, fp, aff, drop_bits, p_drop, N, B, dim, row_n, row_b, tape=None):
    """(sample [R, dim], log q [R]) of the conditional Glow's sampling direction, all layers in one launch (mhe_glow_layers_bf16).
    fp: glow_fused_layout's dict as bf16 (weights) / f32 (biases) device tensors; aff: glow_affine's dict; drop_bits: uint8 [L, 2, R * 64] | None;
    tape: dict(v, y, prm f32 [L, R, 64]; tb, t2, t3 bf16 [L, 2, R, 512]; hf bf16 [L, R, 512]) | None"""
    R = N * B
    _chk(noise, torch.float32, "glow_layers.noise", (R, dim)); _chk(ctab, torch.float32, "glow_layers.ctab")
    L = fp["wxF"].shape[0]
    for k, dt in (("wxF", torch.bfloat16), ("w0F", torch.bfloat16), ("w1F", torch.bfloat16), ("wsF", torch.bfloat16), ("wuF", torch.bfloat16),
                  ("b0", torch.float32), ("b1", torch.float32), ("bs", torch.float32), ("bu", torch.float32)):
        _chk(fp[k], dt, "glow_layers." + k)
    if fp["w0F"].numel() != L * 2 * 512 * 512 or fp["wxF"].numel() != L * 512 * 64 or fp["wsF"].numel() != L * 64 * 512:
        raise _lib.MheError("glow_layers: weight operands do not match hidden 512 / 2 blocks per layer")
    if drop_bits is not None:
        _chk(drop_bits, torch.uint8, "glow_layers.drop_bits")
        if drop_bits.numel() != L * 2 * R * 64:
            raise _lib.MheError(f"glow_layers.drop_bits: {L * 2 * R * 64} bytes expected, got {drop_bits.numel()}")
    out, logq = torch.empty(R, dim, device=noise.device), torch.empty(R, device=noise.device)
    t = tape or {}
    if tape is not None:
        for k, dt, shp in (("v", torch.float32, (L, R, 64)), ("y", torch.float32, (L, R, 64)), ("prm", torch.float32, (L, R, 64)),
                           ("tb", torch.bfloat16, (L, 2, R, 512)), ("t2", torch.bfloat16, (L, 2, R, 512)), ("t3", torch.bfloat16, (L, 2, R, 512)),
                           ("hf", torch.bfloat16, (L, R, 512))):
            _chk(tape[k], dt, "glow_layers.tape." + k, shp)
    check(_lib.lib().mhe_glow_layers_bf16(_ptr(noise), _ptr(ctab), ctab.shape[1], _ptr(fp["wxF"]), _ptr(fp["w0F"]), _ptr(fp["w1F"]), _ptr(fp["wsF"]),
                                          _ptr(fp["wuF"]), _ptr(fp["b0"]), _ptr(fp["b1"]), _ptr(fp["bs"]), _ptr(fp["bu"]), _ptr(aff["AinvT"]),
                                          _ptr(aff["cinv"]), _ptr(aff["const_parts"]), _ptr(drop_bits), float(p_drop), _ptr(out), _ptr(logq),
                                          _ptr(t.get("v")), _ptr(t.get("y")), _ptr(t.get("prm")), _ptr(t.get("tb")), _ptr(t.get("t2")), _ptr(t.get("t3")),
                                          _ptr(t.get("hf")), _ptr(t.get("prmc")), _ptr(t.get("vb")), _ptr(t.get("bits")), N, B, dim, 512, L, 2, row_n, row_b,
                                          _stream()), "mhe_glow_layers_bf16")
    return out, logq


def glow_reverse_chain_supported(R, B, dim, hidden, layers, blocks):
    return bool(_lib.lib().mhe_glow_reverse_chain_supported(R, B, dim, hidden, layers, blocks))


def glow_reverse_chain(g_x, g_logp, q_weight, tape, ctab, fp, aff, p_drop, out, B, dim):
    """the data-gradient chain of the conditional Glow's sampling direction in one launch (mhe_glow_reverse_chain_bf16) over the tape of
    glow_layers (v, prmc, t3, bits).  out: dict(gv f32 [L,R,64]; gpc bf16 [L,R,128]; gt3, gt2 bf16 [L,2,R,512]; gh0 bf16 [L,R,512];
    gct f32 [B,cs]; bsum f32 [B, L*2*2*512]; bfsum f32 [B, L*128]) - all written"""
    R = g_x.shape[0]
    L = tape["v"].shape[0]
    _chk(g_x, torch.float32, "glow_reverse_chain.g_x", (R, dim))
    for k, dt, shp in (("v", torch.float32, (L, R, 64)), ("prmc", torch.float32, (L, R, 128)), ("t3", torch.bfloat16, (L, 2, R, 512))):
        _chk(tape[k], dt, "glow_reverse_chain.tape." + k, shp)
    _chk(tape["bits"], torch.int32, "glow_reverse_chain.tape.bits", (L, 2, 2, B, 512, 2))
    for k, dt, shp in (("gv", torch.float32, (L, R, 64)), ("gpc", torch.bfloat16, (L, R, 128)), ("gt3", torch.bfloat16, (L, 2, R, 512)),
                       ("gt2", torch.bfloat16, (L, 2, R, 512)), ("gh0", torch.bfloat16, (L, R, 512)), ("gct", torch.float32, (B, ctab.shape[1])),
                       ("bsum", torch.float32, (B, L * 2 * 2 * 512)), ("bfsum", torch.float32, (B, L * 128))):
        _chk(out[k], dt, "glow_reverse_chain.out." + k, shp)
    for k in ("wsT", "wuT", "w1T", "w0T", "wxT"):
        _chk(fp[k], torch.bfloat16, "glow_reverse_chain." + k)
    check(_lib.lib().mhe_glow_reverse_chain_bf16(_ptr(g_x), _ptr(g_logp), float(q_weight), _ptr(tape["v"]), _ptr(tape["prmc"]), _ptr(tape["t3"]),
                                                 _ptr(tape["bits"]), _ptr(ctab), ctab.shape[1], _ptr(fp["wsT"]), _ptr(fp["wuT"]), _ptr(fp["w1T"]),
                                                 _ptr(fp["w0T"]), _ptr(fp["wxT"]), _ptr(aff["Ainv"]), float(p_drop), _ptr(out["gv"]), _ptr(out["gpc"]),
                                                 _ptr(out["gt3"]), _ptr(out["gt2"]), _ptr(out["gh0"]), _ptr(out["gct"]), _ptr(out["bsum"]),
                                                 _ptr(out["bfsum"]), R, B, dim, 512, L, 2, _stream()), "mhe_glow_reverse_chain_bf16")
    return out
