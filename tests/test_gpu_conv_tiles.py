"""GPU parity of EVERY instantiation of the implicit-GEMM convolution, the large tiles included.

The launcher picks the 256x256 tiles only for big problems (>= 192 output tiles), which no oracle-sized test reaches
by itself; `mhe_conv_desc.tile` forces a kernel variant per call, so each variant is compared with F.conv2d on the same
bf16-rounded operands (f32 accumulate, tolerance = bf16 output rounding), in all the forms the trunk uses it:
plain + batch statistics, fused output affine / residual / ReLU, producer-BatchNorm operand load, the residual-tail
operand load, and the data-gradient form with ReLU gate and BatchNorm-reverse sums.  Then shapes at which the launcher
selects the large tile by itself.  Replaces torchvision's convolutions: reference hand/network.py:54-61,110."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 6e-3                       # bf16 output rounding (2^-9) relative to the tensor's scale

# variant id -> (name, supports the producer-BatchNorm / residual-tail operand loads)
VARIANTS = {2: ("256x256", True), 3: ("256x128", True), 4: ("256x64", True), 7: ("8-phase 256x256", False), 13: ("8-phase 256x128", False)}
# B, H, W, Cin, Cout, k, stride, pad
SHAPES = [
    (2, 24, 20, 64, 256, 1, 1, 0),      # K = a single 64-deep stage, partial M tile
    (3, 20, 20, 128, 320, 3, 1, 1),     # partial N tile, padding taps
    (2, 33, 31, 256, 512, 1, 2, 0),     # stride-2 1x1 (the downsample form)
    (2, 17, 19, 192, 256, 3, 2, 1),     # stride-2 3x3, odd number of K stages (27)
    (5, 16, 16, 256, 256, 3, 1, 1),     # layer3 conv2 at a small batch: 5 M tiles
]


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()


def _operands(seed, B, H, W, Cin, Cout, k):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).bfloat16().float()
    return g, x, w


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", sorted(VARIANTS), ids=lambda v: VARIANTS[v][0].replace(" ", "-"))
def test_forced_variant_matches_torch(gpu_lib, variant, shape):
    from mhentropy_amd import ops, resnet
    B, H, W, Cin, Cout, k, stride, pad = shape
    g, x, w = _operands(variant * 100 + Cin + k, B, H, W, Cin, Cout, k)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    xd = _nhwc(x)
    wd = resnet.pack_conv_weight(w, torch.bfloat16).cuda()
    tile = variant + 1
    stats = ops.stat_unit(Cout, "cuda")
    y = ops.conv2d_nhwc(xd, wd, k, k, stride, pad, stats=stats, tile=tile)
    assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, TOL, what="raw conv")
    n = ref.numel() / Cout
    st = ops.stat_totals(stats).cpu()
    ys = y.double().cpu().permute(0, 3, 1, 2)           # the statistics are those of the output AS STORED (what the consumer normalises)
    assert_close(st[0] / n, ys.mean((0, 2, 3)), 1e-5, 1e-5, what="batch mean")
    assert_close(st[1] / n, (ys ** 2).mean((0, 2, 3)), 1e-5, what="batch E[x^2]")
    assert (stats[0].abs().sum((1, 2)) > 0).sum().item() >= min(ops.stat_shards(), (ref.numel() // Cout + 255) // 256), \
        "statistics must be spread over the shards"
    # fused eval-mode epilogue: relu(conv*scale+shift + residual)
    osc = torch.rand(Cout, generator=g) + 0.5
    osh = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(ref.shape, generator=g).bfloat16().float()
    y2 = ops.conv2d_nhwc(xd, wd, k, k, stride, pad, out_scale=osc.cuda(), out_shift=osh.cuda(), residual=_nhwc(res),
                         relu_out=True, tile=tile)
    ref2 = torch.relu(ref * osc.view(1, -1, 1, 1) + osh.view(1, -1, 1, 1) + res)
    assert_close(y2.float().cpu().permute(0, 3, 1, 2), ref2, TOL, what="fused epilogue")
    if VARIANTS[variant][1]:
        # producer BatchNorm + ReLU applied while the operand is loaded (padding must stay zero)
        isc = torch.rand(Cin, generator=g) + 0.5
        ish = torch.randn(Cin, generator=g) * 0.3
        y3 = ops.conv2d_nhwc(xd, wd, k, k, stride, pad, in_scale=isc.cuda(), in_shift=ish.cuda(), relu_in=True, tile=tile)
        xin = torch.relu(x * isc.view(1, -1, 1, 1) + ish.view(1, -1, 1, 1)).bfloat16().float()
        ref3 = F.conv2d(xin.double(), w.double(), None, stride, pad)
        assert_close(y3.float().cpu().permute(0, 3, 1, 2), ref3, TOL, what="fused input transform")


@pytest.mark.parametrize("shape", [(2, 24, 20, 256, 256, 1, 1, 0), (3, 20, 20, 128, 320, 3, 1, 1), (5, 16, 16, 256, 256, 3, 1, 1)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", [1, 2, 7, 13], ids=lambda v: {1: "128x128"}.get(v, VARIANTS.get(v, ("",))[0]).replace(" ", "-"))
def test_data_gradient_form_gate_and_bn_sums(gpu_lib, variant, shape):
    """y = (conv + residual) [mask > 0] and, for two BatchNorm units fed by y, sum y and sum y (bn_y - mean) invstd
    (mhe_conv2d_masked_nhwc) against the same quantities from torch; reference hand/CrossModalHand.py:455-470 (backward)."""
    from mhentropy_amd import ops, resnet
    B, H, W, Cin, Cout, k, stride, pad = shape
    g, x, w = _operands(variant * 10 + Cout + k, B, H, W, Cin, Cout, k)
    conv = F.conv2d(x.double(), w.double(), None, stride, pad)
    res = torch.randn(conv.shape, generator=g).bfloat16().float()
    mask = torch.randn(conv.shape, generator=g).bfloat16().float()
    bny = [(torch.randn(conv.shape, generator=g) * 1.5 + 0.3).bfloat16().float() for _ in range(2)]
    mi = [torch.stack([torch.randn(Cout, generator=g) * 0.2, torch.rand(Cout, generator=g) + 0.5]) for _ in range(2)]
    want = (conv + res) * (mask > 0)
    st = [ops.stat_unit(Cout, "cuda") for _ in range(2)]
    y = ops.conv2d_nhwc(_nhwc(x), resnet.pack_conv_weight(w, torch.bfloat16).cuda(), k, k, stride, pad, residual=_nhwc(res),
                        mask=_nhwc(mask), bn=[(_nhwc(bny[u]), mi[u].cuda().contiguous(), st[u]) for u in range(2)], tile=variant + 1)
    assert_close(y.float().cpu().permute(0, 3, 1, 2), want, TOL, what="gated data gradient")
    # the sums are taken over the tile as staged for the store (convolution rounded to bf16, then + residual, gate), exactly
    # what the separate reduce pass would read back: each term carries a rounding error of <= 2^-9 of its value, so a
    # channel's sum may differ from the exact one by a few 2^-9 sqrt(sum of squares)
    for u in range(2):
        s = ops.stat_totals(st[u]).cpu()
        xhat = (bny[u].double() - mi[u][0].view(1, -1, 1, 1)) * mi[u][1].view(1, -1, 1, 1)
        for name, got, terms in (("sum g", s[0], conv * (mask > 0)), ("sum g xhat", s[1], conv * (mask > 0) * xhat)):
            exact = (want if name == "sum g" else want * xhat).sum((0, 2, 3))
            bound = 5 * 2.0 ** -9 * terms.pow(2).sum((0, 2, 3)).sqrt() + 1e-4 * exact.abs().max()
            worst = ((got - exact).abs() / bound).max().item()
            assert worst <= 1.0, f"{name} (unit {u}): |diff| is {worst:.2f} x the bf16 staging bound"


@pytest.mark.parametrize("variant", [1, 2], ids=["128x128", "256x256"])
@pytest.mark.parametrize("affine2", [False, True], ids=["identity", "downsample-bn"])
def test_residual_tail_operand_load(gpu_lib, variant, affine2):
    """mhe_conv1x1_residual_in_nhwc: a = relu(x*s+t + (x2*s2+t2 | x2)), y = conv1x1(a), a written out once."""
    from mhentropy_amd import ops, resnet
    B, H, W, Cin, Cout = 3, 20, 24, 256, 320
    g, x, w = _operands(5 + variant, B, H, W, Cin, Cout, 1)
    x2 = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    sc2, sh2 = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    a = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + (x2 * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1) if affine2 else x2)
    a = torch.relu(a).bfloat16().float()
    ref = F.conv2d(a.double(), w.double())
    a_out = torch.empty(B, H, W, Cin, device="cuda", dtype=torch.bfloat16)
    stats = ops.stat_unit(Cout, "cuda")
    y = ops.conv1x1_residual_in(_nhwc(x), _nhwc(x2), resnet.pack_conv_weight(w, torch.bfloat16).cuda(), sc.cuda(), sh.cuda(),
                                sc2.cuda() if affine2 else None, sh2.cuda() if affine2 else None, a_out=a_out, stats=stats,
                                tile=variant + 1)
    assert_close(a_out.float().cpu().permute(0, 3, 1, 2), a, 4e-3, what="block output written by the operand load")
    assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, TOL, what="conv1x1 of the fused tail")
    n = ref.numel() / Cout
    assert_close(ops.stat_totals(stats).cpu()[0] / n, y.double().cpu().permute(0, 3, 1, 2).mean((0, 2, 3)), 1e-5, 1e-5, what="batch mean (of the stored output)")


# the last three: more tiles than the 256 persistent workgroups (a second tile per workgroup, its first K tiles loaded during the first one's
# epilogue) with one and with two column tiles, and a pixel-tile count that is not a multiple of 8 (plain tile order)
@pytest.mark.parametrize("geom", [(4, 16, 16, 1024, 256), (2, 8, 16, 2048, 512), (8, 16, 16, 512, 256), (36, 32, 32, 128, 256), (24, 16, 48, 128, 512),
                                  (2, 24, 24, 256, 512), (8, 16, 16, 2048, 512), (40, 16, 16, 1024, 256)], ids=lambda g: "x".join(map(str, g)))
@pytest.mark.parametrize("affine2", [False, True], ids=["identity", "downsample-bn"])
def test_residual_tail_kernel_with_transfer_waves(gpu_lib, geom, affine2):
    """variant 10 (csrc/conv_tail.hip): the residual tail + conv1 of the wide layers on a 128 x 256 tile, load / transform / store work in
    waves of their own.  Against torch in f64, and against the 128x128 variant: the block output it writes must be the same to the bit
    (same arithmetic in the same order), the products agree to accumulation order."""
    from mhentropy_amd import ops, resnet
    B, H, W, Cin, Cout = geom
    assert ops.conv_tile_choice(B, H, W, Cin, Cout, 1, 1, 0, torch.bfloat16, 2) == 10
    g, x, w = _operands(31 + Cin // 512, B, H, W, Cin, Cout, 1)
    x2 = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    sc2, sh2 = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    a = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + (x2 * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1) if affine2 else x2)
    a = torch.relu(a).bfloat16().float()
    ref = F.conv2d(a.double(), w.double())
    wp = resnet.pack_conv_weight(w, torch.bfloat16).cuda()
    out = {}
    for tile in (11, 2):
        a_out = torch.full((B, H, W, Cin), float("nan"), device="cuda", dtype=torch.bfloat16)
        stats = ops.stat_unit(Cout, "cuda")
        y = ops.conv1x1_residual_in(_nhwc(x), _nhwc(x2), wp, sc.cuda(), sh.cuda(), sc2.cuda() if affine2 else None,
                                    sh2.cuda() if affine2 else None, a_out=a_out, stats=stats, tile=tile)
        out[tile] = (y, a_out, stats)
    y, a_out, stats = out[11]
    assert torch.equal(a_out, out[2][1]), "block output differs from the 128x128 variant's"
    assert_close(a_out.float().cpu().permute(0, 3, 1, 2), a, 4e-3, what="block output written by the transfer waves")
    assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, TOL, what="conv1x1 of the fused tail")
    assert_close(y.float().cpu(), out[2][0].float().cpu(), 8e-3, what="against the 128x128 variant")
    n = ref.numel() / Cout
    yd = y.double().cpu().permute(0, 3, 1, 2)
    assert_close(ops.stat_totals(stats).cpu()[0] / n, yd.mean((0, 2, 3)), 1e-5, 1e-5, what="batch mean (of the stored output)")
    assert_close(ops.stat_totals(stats).cpu()[1] / n, (yd * yd).mean((0, 2, 3)), 1e-5, 1e-5, what="batch mean square (of the stored output)")
    # without a_out / statistics
    for tile in (16, 11, 0):
        y2 = ops.conv1x1_residual_in(_nhwc(x), _nhwc(x2), wp, sc.cuda(), sh.cuda(), sc2.cuda() if affine2 else None,
                                     sh2.cuda() if affine2 else None, tile=tile)
        assert torch.equal(y2, y), tile


@pytest.mark.parametrize("shape", [(12, 64, 64, 64, 256, 1, 1, 0), (192, 16, 16, 256, 256, 3, 1, 1), (48, 32, 32, 128, 512, 1, 1, 0),
                                   (192, 8, 8, 512, 512, 3, 1, 1), (192, 16, 16, 512, 512, 3, 2, 1)],
                         ids=lambda s: "x".join(map(str, s)))
def test_launcher_selected_large_tile_matches_torch(gpu_lib, shape):
    """shapes big enough (>= 192 output tiles) that the launcher itself takes a 256-row variant, as at the bench batch: 256x256 (2, 7) or,
    for the 3x3 layers with 128 output channels / too few pixels for 256-channel tiles, the phase-pipelined kernel on 256x128 (13)"""
    from mhentropy_amd import ops, resnet, _lib
    import ctypes as C
    B, H, W, Cin, Cout, k, stride, pad = shape
    d = _lib.ConvDesc(B, H, W, Cin, Cout, k, k, stride, pad, ops.BF16, 0, 0, 0)
    want = (13,) if (k == 3 and B * H * W // (stride * stride) < 49152) else (2, 7)
    assert _lib.lib().mhe_conv_tile(C.byref(d)) in want, f"expected variant {want} for this geometry, got {_lib.lib().mhe_conv_tile(C.byref(d))}"
    g, x, w = _operands(Cin + Cout, B, H, W, Cin, Cout, k)
    ref = F.conv2d(x, w, None, stride, pad)              # f32 on the bf16-rounded operands (29 GMAC at the 3x3 shape)
    stats = ops.stat_unit(Cout, "cuda")
    y = ops.conv2d_nhwc(_nhwc(x), resnet.pack_conv_weight(w, torch.bfloat16).cuda(), k, k, stride, pad, stats=stats)
    assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, TOL, what="raw conv")
    n = ref.numel() / Cout
    st = ops.stat_totals(stats).cpu()
    ys = y.double().cpu().permute(0, 3, 1, 2)           # statistics of the output as stored
    assert_close(st[0] / n, ys.mean((0, 2, 3)), 1e-5, 1e-5, what="batch mean")
    assert_close(st[1] / n, (ys ** 2).mean((0, 2, 3)), 1e-5, what="batch E[x^2]")
    assert (stats[0].abs().sum((1, 2)) > 0).sum().item() > 1, "statistics must be spread over more than one shard"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_statistics_use_every_shard(gpu_lib, dtype):
    """regression: the stem once added every workgroup's partial sums into shard 0 (contended atomics, 3x slower)"""
    from mhentropy_amd import ops, resnet
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 3, 128, 128, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    stats = ops.stat_unit(64, "cuda")
    ops.stem_conv7x7s2(x.cuda(), resnet.pack_stem_weight(w, dtype).cuda(), dtype, stats=stats)
    assert (stats[0].abs().sum((1, 2)) > 0).all(), "every statistic shard should receive workgroups"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("Cin,Cout", [(256, 64), (512, 128), (1024, 256)])
def test_dgrad_with_batchnorm_reverse_applied_on_load(gpu_lib, Cin, Cout, dt):
    """mhe_conv1x1_residual_in_masked_nhwc as the train step uses it: operand gy = k2 g + k1 y + k0 formed in the load (and written
    to a_out), 1x1 product with the data-gradient weights, ReLU gate and BatchNorm-reverse sums in the epilogue - against the two-pass
    form (mhe_bn_bwd_apply_nhwc, then mhe_conv2d_masked_nhwc) and against torch on the rounded operands"""
    from mhentropy_amd import ops
    B, H = 3, 20                                   # M = 1200: partial tiles
    gen = torch.Generator().manual_seed(Cin + Cout)
    rnd = lambda *s: torch.randn(*s, generator=gen)
    g, y = rnd(B, H, H, Cin).to(dt).cuda(), (rnd(B, H, H, Cin) * 2 + 0.3).to(dt).cuda()
    coef = torch.stack([torch.rand(Cin, generator=gen) + 0.5, rnd(Cin) * 0.2, rnd(Cin) * 0.1]).cuda().contiguous()
    w = (rnd(Cout, Cin) / Cin ** 0.5).to(dt).cuda().contiguous()
    mask = rnd(B, H, H, Cout).to(dt).cuda()
    bn_y = rnd(B, H, H, Cout).to(dt).cuda()
    mi = torch.stack([rnd(Cout) * 0.1, torch.rand(Cout, generator=gen) + 0.5]).cuda().contiguous()
    st1, st2 = (ops.stat_unit(Cout, "cuda") for _ in range(2))
    gy = torch.empty_like(g)
    out = ops.conv1x1_dgrad_bn_apply(g, y, coef, w, gy, mask, bn=[(bn_y, mi, st1)])
    gy_ref = (coef[0] * g.float() + coef[1] * y.float() + coef[2])
    tol = 1e-5 if dt == torch.float32 else 8e-3
    assert_close(gy.float().cpu(), gy_ref.cpu(), tol, what="operand written out")
    want = (gy.float() @ w.float().t()) * (mask.float() > 0)          # from the operand as stored (bf16: its rounded value feeds the product)
    assert_close(out.float().cpu(), want.cpu(), 2e-5 if dt == torch.float32 else 8e-3, what="gated data gradient")
    ref = ops.conv2d_nhwc(gy, w, 1, 1, 1, 0, mask=mask, bn=[(bn_y, mi, st2)])
    assert_close(out.float().cpu(), ref.float().cpu(), 2e-5 if dt == torch.float32 else 8e-3, what="vs the two-pass form")
    s1, s2 = ops.stat_totals(st1).cpu(), ops.stat_totals(st2).cpu()
    scale = s2.abs().max(1, keepdim=True)[0] + 1.0
    assert ((s1 - s2).abs() <= (1e-4 if dt == torch.float32 else 3e-2) * scale).all(), "BatchNorm-reverse sums"


@pytest.mark.parametrize("geom", [(8, 16, 1024, 256), (4, 16, 2048, 512), (24, 32, 256, 512)], ids=lambda g: "x".join(map(str, g)))
def test_dgrad_with_batchnorm_reverse_on_load_transfer_wave_kernel(gpu_lib, geom):
    """the data-gradient form of variant 10 (csrc/conv_tail.hip), as the train step uses it for conv3 of layer3 / layer4: against the 128x128
    variant (operand written out: to the bit; gated gradient and BatchNorm-reverse sums: to accumulation order); the third geometry has more
    tiles than persistent workgroups"""
    from mhentropy_amd import ops
    B, H, Cin, Cout = geom
    dt = torch.bfloat16
    assert ops.conv_tile_choice(B, H, H, Cin, Cout, 1, 1, 0, dt, 2) == 10
    gen = torch.Generator().manual_seed(Cin + Cout)
    rnd = lambda *s: torch.randn(*s, generator=gen)
    g, y = rnd(B, H, H, Cin).to(dt).cuda(), (rnd(B, H, H, Cin) * 2 + 0.3).to(dt).cuda()
    coef = torch.stack([torch.rand(Cin, generator=gen) + 0.5, rnd(Cin) * 0.2, rnd(Cin) * 0.1]).cuda().contiguous()
    w = (rnd(Cout, Cin) / Cin ** 0.5).to(dt).cuda().contiguous()
    mask = rnd(B, H, H, Cout).to(dt).cuda()
    bn_y = rnd(B, H, H, Cout).to(dt).cuda()
    mi = torch.stack([rnd(Cout) * 0.1, torch.rand(Cout, generator=gen) + 0.5]).cuda().contiguous()
    res = {}
    for tile in (11, 2):
        st = ops.stat_unit(Cout, "cuda")
        gy = torch.full_like(g, float("nan"))
        out = ops.conv1x1_dgrad_bn_apply(g, y, coef, w, gy, mask, bn=[(bn_y, mi, st)], tile=tile)
        res[tile] = (out, gy, ops.stat_totals(st).cpu())
    out, gy, s1 = res[11]
    assert torch.equal(gy, res[2][1]), "operand written out differs from the 128x128 variant's"
    want = (gy.float() @ w.float().t()) * (mask.float() > 0)
    assert_close(out.float().cpu(), want.cpu(), 8e-3, what="gated data gradient")
    assert_close(out.float().cpu(), res[2][0].float().cpu(), 8e-3, what="vs the 128x128 variant")
    s2 = res[2][2]
    scale = s2.abs().max(1, keepdim=True)[0] + 1.0
    assert ((s1 - s2).abs() <= 3e-2 * scale).all(), "BatchNorm-reverse sums"


@pytest.mark.parametrize("geom", [(64, 16, 256, 1024), (64, 8, 512, 2048), (128, 16, 512, 256)], ids=lambda g: "x".join(map(str, g)))
@pytest.mark.parametrize("bn_load", [False, True], ids=["plain", "bn-on-load"])
def test_resident_slab_kernel_with_transfer_waves(gpu_lib, geom, bn_load):
    """variant 11 (csrc/conv_wide.hip): conv3 of layer3 / layer4 (and 512 -> 256, four slabs) with the weight slab resident in LDS and the
    load / (BatchNorm +) store work in transfer waves; against matmul on the rounded operand, the phase-pipelined / register-staged variant,
    and the batch statistics of the stored outputs"""
    from mhentropy_amd import ops
    B, H, Cin, Cout = geom
    dt = torch.bfloat16
    assert (ops.conv_tile_choice(B, H, H, Cin, Cout, 1, 1, 0, dt, int(bn_load)) == 11) == (Cin == 256)     # 512 input channels: on request only
    gen = torch.Generator().manual_seed(Cin + Cout + bn_load)
    x = torch.randn(B, H, H, Cin, generator=gen).to(dt).cuda()
    w = (torch.randn(Cout, Cin, generator=gen) / Cin ** 0.5).to(dt).cuda().contiguous()
    sc = (torch.rand(Cin, generator=gen) + 0.5).cuda() if bn_load else None
    sh = (torch.randn(Cin, generator=gen) * 0.3).cuda() if bn_load else None
    a = torch.relu(x.float() * sc + sh).to(dt).float() if bn_load else x.float()
    want = a.reshape(-1, Cin) @ w.float().t()
    res = {}
    for tile in (12, 2):
        st = ops.stat_unit(Cout, "cuda")
        y = ops.conv2d_nhwc(x, w, 1, 1, 1, 0, in_scale=sc, in_shift=sh, relu_in=bn_load, stats=st, tile=tile)
        res[tile] = (y, ops.stat_totals(st).cpu())
    y, st = res[12]
    assert_close(y.float().cpu().reshape(-1, Cout), want.cpu(), TOL, what="1x1 product")
    assert_close(y.float().cpu(), res[2][0].float().cpu(), 8e-3, what="vs the 128x128 variant")
    n = y.numel() / Cout
    yd = y.double().cpu().reshape(-1, Cout)
    assert_close(st[0] / n, yd.mean(0), 1e-5, 1e-5, what="batch mean (of the stored output)")
    assert_close(st[1] / n, (yd * yd).mean(0), 1e-5, 1e-5, what="batch mean square (of the stored output)")
    y2 = ops.conv2d_nhwc(x, w, 1, 1, 1, 0, in_scale=sc, in_shift=sh, relu_in=bn_load, tile=12)       # without statistics
    assert torch.equal(y2, y)


@pytest.mark.parametrize("tile", [0, 2, 8])
@pytest.mark.parametrize("masked", [True, False])
def test_half_resolution_residual(gpu_lib, tile, masked):
    """mhe_conv_desc.res_half: a residual on the coarse grid [B, H/2, W/2, C] added at the even output positions only (the gradient of
    a stride-2 1x1 shortcut joining the main branch) == the same launch fed the scattered full-resolution tensor; odd sizes included"""
    from mhentropy_amd import ops
    B, H, W, Cin, Cout = 3, 14, 10, 64, 256
    gen = torch.Generator().manual_seed(tile + 7 * masked)
    bf = torch.bfloat16
    x = torch.randn(B, H, W, Cin, generator=gen).to(bf).cuda()
    w = (torch.randn(Cout, Cin, generator=gen) / 8).to(bf).cuda().contiguous()
    half = torch.randn(B, (H + 1) // 2, (W + 1) // 2, Cout, generator=gen).to(bf).cuda()
    mask = torch.randn(B, H, W, Cout, generator=gen).to(bf).cuda() if masked else None
    full = ops.upsample2(half, H, W)
    bn1 = bn2 = None
    if masked:
        bn_y = torch.randn(B, H, W, Cout, generator=gen).to(bf).cuda()
        mi = torch.stack([torch.randn(Cout, generator=gen) * 0.1, torch.rand(Cout, generator=gen) + 0.5]).cuda().contiguous()
        st1, st2 = (ops.stat_unit(Cout, "cuda") for _ in range(2))
        bn1, bn2 = [(bn_y, mi, st1)], [(bn_y, mi, st2)]
    got = ops.conv2d_nhwc(x, w, 1, 1, 1, 0, residual=half, mask=mask, bn=bn1, tile=tile, res_half=True)
    want = ops.conv2d_nhwc(x, w, 1, 1, 1, 0, residual=full, mask=mask, bn=bn2, tile=tile)
    assert torch.equal(got, want)
    if masked:
        assert_close(ops.stat_totals(st1).cpu(), ops.stat_totals(st2).cpu(), 1e-6, 1e-5, what="BatchNorm-reverse sums")


@pytest.mark.parametrize("bn_load", [False, True], ids=["plain", "bn-on-load"])
@pytest.mark.parametrize("shape", [(16, 64, 64, 64, 256), (18, 61, 60, 64, 256), (64, 32, 32, 128, 512), (17, 64, 61, 128, 256), (17, 64, 61, 256, 384),
                                   (16, 64, 64, 64, 64), (18, 61, 60, 64, 64)],
                         ids=lambda s: "x".join(map(str, s)))
def test_streaming_1x1_kernel(gpu_lib, shape, bn_load):
    """conv1x1_stream_kernel (variant 8: K = 64 / 128 input channels, weights resident in LDS, 64-pixel chunks): the launcher picks it
    for these shapes by itself AND forced; against conv2d on the same bf16-rounded operands (with the producer BatchNorm + ReLU applied
    to the operand in f32 and rounded, as the kernel stores it), statistics of the output as stored; pixel counts that are not a
    multiple of the chunk included"""
    from mhentropy_amd import ops, resnet, _lib
    import ctypes as C
    B, H, W, Cin, Cout = shape
    if Cin == 256:
        pytest.skip("K = 256 takes the streaming kernel in its data-gradient form only (the plain form is faster on the tiled kernel)")
    g, x, w = _operands(Cin + Cout + H, B, H, W, Cin, Cout, 1)
    xd = _nhwc(x)
    wd = resnet.pack_conv_weight(w, torch.bfloat16).cuda()
    kw = {}
    xin = x
    if bn_load:
        sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
        kw = dict(in_scale=sc.cuda(), in_shift=sh.cuda(), relu_in=True)
        xin = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).bfloat16().float()
    ref = F.conv2d(xin.double(), w.double())
    d = _lib.ConvDesc(B, H, W, Cin, Cout, 1, 1, 1, 0, ops.BF16, 0, 0, 0)
    assert _lib.lib().mhe_conv_tile_mode(C.byref(d), 1 if bn_load else 0) == 8, "the launcher should pick the streaming kernel here"
    for tile in (0, 9):
        stats = ops.stat_unit(Cout, "cuda")
        y = ops.conv2d_nhwc(xd, wd, 1, 1, 1, 0, stats=stats, tile=tile, **kw)
        assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, TOL, what="raw conv")
        n = ref.numel() / Cout
        st = ops.stat_totals(stats).cpu()
        ys = y.double().cpu().permute(0, 3, 1, 2)
        assert_close(st[0] / n, ys.mean((0, 2, 3)), 1e-5, 1e-5, what="batch mean")
        assert_close(st[1] / n, (ys ** 2).mean((0, 2, 3)), 1e-5, what="batch E[x^2]")
        assert (stats[0].abs().sum((1, 2)) > 0).sum().item() > 8, "statistics must be spread over the shards"
    y2 = ops.conv2d_nhwc(xd, wd, 1, 1, 1, 0, tile=9, **kw)                  # without statistics
    assert torch.equal(y2, y)


@pytest.mark.parametrize("nbn,res", [(0, "none"), (1, "full"), (2, "full"), (1, "half")])
@pytest.mark.parametrize("shape", [(18, 61, 60, 64, 256), (64, 32, 32, 128, 512), (17, 64, 61, 256, 384)], ids=lambda s: "x".join(map(str, s)))
def test_streaming_1x1_kernel_data_gradient_form(gpu_lib, shape, nbn, res):
    """the streaming kernel with the data-gradient epilogue (ReLU gate, residual at full / half resolution, BatchNorm-reverse sums of up to
    two units): same results as the register-staged 128x128 data-gradient kernel on the same operands"""
    from mhentropy_amd import ops, resnet
    B, H, W, Cin, Cout = shape
    g, x, w = _operands(Cin + nbn + H, B, H, W, Cin, Cout, 1)
    xd = _nhwc(x)
    wd = resnet.pack_conv_weight(w, torch.bfloat16).cuda()
    rnd = lambda *s: torch.randn(*s, generator=g).bfloat16().cuda()
    mask = rnd(B, H, W, Cout)
    residual = None if res == "none" else (rnd(B, H, W, Cout) if res == "full" else rnd(B, (H + 1) // 2, (W + 1) // 2, Cout))
    bns = [[], []]
    for k in range(2):
        for u in range(nbn):
            by = rnd(B, H, W, Cout) if k == 0 else bns[0][u][0]
            mi = (torch.stack([torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5]).cuda().contiguous() if k == 0 else bns[0][u][1])
            bns[k].append((by, mi, ops.stat_unit(Cout, "cuda")))
    kw = dict(residual=residual, mask=mask, res_half=res == "half")
    got = ops.conv2d_nhwc(xd, wd, 1, 1, 1, 0, bn=bns[0] or None, tile=9, **kw)
    auto = ops.conv2d_nhwc(xd, wd, 1, 1, 1, 0, tile=0, **kw)
    want = ops.conv2d_nhwc(xd, wd, 1, 1, 1, 0, bn=bns[1] or None, tile=2, **kw)
    assert torch.equal(got, auto), "the launcher picks the streaming kernel for this shape"
    assert_close(got.float().cpu(), want.float().cpu(), 8e-3, what="gated data gradient")        # different summation order before the bf16 rounding
    for u in range(nbn):
        a, b = ops.stat_totals(bns[0][u][2]).cpu(), ops.stat_totals(bns[1][u][2]).cpu()
        scale = b.abs().max(1, keepdim=True)[0] + 1.0
        assert ((a - b).abs() <= 2e-2 * scale).all(), f"BatchNorm-reverse sums of unit {u}"


@pytest.mark.parametrize("form", ["plain", "bn-on-load", "data-gradient"])
# (300, 17): more images than the 256 workgroups - the flat pair stream crosses image boundaries inside a workgroup, odd height
@pytest.mark.parametrize("B,H", [(128, 16), (130, 19), (300, 17)])
def test_row_streaming_3x3_kernel(gpu_lib, B, H, form):
    """conv3x3_c64_stream_kernel (variant 9: 3x3 / stride 1 / pad 1, 64 -> 64 channels, 64-pixel-wide maps; weights and a ring of input
    rows resident in LDS): picked by the launcher and forced; against conv2d on the bf16-rounded operands (producer BatchNorm + ReLU
    applied in f32 to the operand and rounded, zero padding AFTER it), statistics of the stored output, and - data-gradient form - against
    the register-staged data-gradient kernel"""
    from mhentropy_amd import ops, resnet
    W, C = 64, 64
    g, x, w = _operands(B + H, B, H, W, C, C, 3)
    xd = _nhwc(x)
    wd = resnet.pack_conv_weight(w, torch.bfloat16).cuda()
    if form == "data-gradient":
        rnd = lambda *s: torch.randn(*s, generator=g).bfloat16().cuda()
        mask, res, by = rnd(B, H, W, C), rnd(B, H, W, C), rnd(B, H, W, C)
        mi = torch.stack([torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5]).cuda().contiguous()
        st = [ops.stat_unit(C, "cuda") for _ in range(2)]
        got = ops.conv2d_nhwc(xd, wd, 3, 3, 1, 1, residual=res, mask=mask, bn=[(by, mi, st[0])], tile=10)
        auto = ops.conv2d_nhwc(xd, wd, 3, 3, 1, 1, residual=res, mask=mask, tile=0)
        want = ops.conv2d_nhwc(xd, wd, 3, 3, 1, 1, residual=res, mask=mask, bn=[(by, mi, st[1])], tile=1)
        assert torch.equal(got, auto)
        assert_close(got.float().cpu(), want.float().cpu(), 8e-3, what="gated data gradient")
        a, b = ops.stat_totals(st[0]).cpu(), ops.stat_totals(st[1]).cpu()
        assert ((a - b).abs() <= 2e-2 * (b.abs().max(1, keepdim=True)[0] + 1.0)).all(), "BatchNorm-reverse sums"
        return
    kw, xin = {}, x
    if form == "bn-on-load":
        sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
        kw = dict(in_scale=sc.cuda(), in_shift=sh.cuda(), relu_in=True)
        xin = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).bfloat16().float()
    ref = F.conv2d(xin.double(), w.double(), None, 1, 1)
    for tile in (0, 10):
        stats = ops.stat_unit(C, "cuda")
        y = ops.conv2d_nhwc(xd, wd, 3, 3, 1, 1, stats=stats, tile=tile, **kw)
        assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, TOL, what="raw conv")
        n = ref.numel() / C
        stt = ops.stat_totals(stats).cpu()
        ys = y.double().cpu().permute(0, 3, 1, 2)
        assert_close(stt[0] / n, ys.mean((0, 2, 3)), 1e-5, 1e-5, what="batch mean")
        assert_close(stt[1] / n, (ys ** 2).mean((0, 2, 3)), 1e-5, what="batch E[x^2]")
    forced = ops.conv2d_nhwc(xd, wd, 3, 3, 1, 1, tile=1, **kw)
    assert_close(y.float().cpu(), forced.float().cpu(), 8e-3, what="vs the tiled kernel")


@pytest.mark.parametrize("geom", [(4, 16, 16, 64, 64), (6, 16, 16, 64, 128), (3, 16, 16, 128, 128), (2, 8, 8, 128, 64), (40, 32, 32, 64, 64), (80, 16, 16, 128, 128)],
                         ids=lambda g: "x".join(map(str, g)))
@pytest.mark.parametrize("affine2", [False, True], ids=["identity", "downsample-bn"])
def test_bottleneck_tail_with_conv3_reevaluated(gpu_lib, geom, affine2):
    """variant 12 (csrc/conv_fuse.hip): statistics-only conv3 + the fused tail that evaluates conv3 again, against the path it replaces -
    conv3 written out (mhe_conv2d_nhwc, BatchNorm on load) then read back by mhe_conv1x1_residual_in_nhwc.  Same products in the same
    order: block output, next conv1 output bit-identical; statistics to the summation order of the per-thread partials.  And against torch in f64 (bf16 tolerance).
    The last two geometries give every workgroup several tiles (persistent loop, cross-tile prefetch)."""
    from mhentropy_amd import ops, resnet
    B, H, W, Cb, N2 = geom
    C4 = 4 * Cb
    g = torch.Generator().manual_seed(Cb + N2 + B)
    y2 = torch.randn(B, Cb, H, W, generator=g).bfloat16().float()
    idt = torch.randn(B, C4, H, W, generator=g).bfloat16().float()
    w3 = (torch.randn(C4, Cb, 1, 1, generator=g) * (2.0 / Cb) ** 0.5).bfloat16().float()
    w1 = (torch.randn(N2, C4, 1, 1, generator=g) * (2.0 / C4) ** 0.5).bfloat16().float()
    s2, h2 = torch.rand(Cb, generator=g) + 0.5, torch.randn(Cb, generator=g) * 0.3
    s3, h3 = torch.rand(C4, generator=g) + 0.5, torch.randn(C4, generator=g) * 0.3
    si, hi = (torch.rand(C4, generator=g) + 0.5, torch.randn(C4, generator=g) * 0.3) if affine2 else (None, None)
    cu = lambda t: None if t is None else t.cuda()
    y2d, idd = _nhwc(y2), _nhwc(idt)
    w3d, w1d = resnet.pack_conv_weight(w3, torch.bfloat16).cuda(), resnet.pack_conv_weight(w1, torch.bfloat16).cuda()
    S = ops.stat_shards()
    # the unfused path: conv3 with bn2 + relu on the operand load (+ statistics), then the MODE 2 tail
    st3_ref, st1_ref = ops.stat_unit(C4, "cuda"), ops.stat_unit(N2, "cuda")
    y3 = ops.conv2d_nhwc(y2d, w3d, 1, 1, 1, 0, in_scale=cu(s2), in_shift=cu(h2), relu_in=True, stats=st3_ref)
    a_ref = torch.empty_like(y3)
    y1_ref = ops.conv1x1_residual_in(y3, idd, w1d, cu(s3), cu(h3), cu(si), cu(hi), a_out=a_ref, stats=st1_ref)
    # statistics-only conv3: same sums, nothing stored
    st3 = ops.stat_unit(C4, "cuda")
    ops.conv1x1_stats(y2d, w3d, cu(s2), cu(h2), st3)
    assert_close(ops.stat_totals(st3).cpu(), ops.stat_totals(st3_ref).cpu(), 1e-6, what="conv3 statistics without the store")
    # fused tail
    assert ops.bottleneck_tail_supported(B, H, W, Cb, N2)
    st1 = ops.stat_unit(N2, "cuda")
    a, y1 = ops.bottleneck_tail(y2d, (cu(s2), cu(h2)), w3d, (cu(s3), cu(h3)), idd, (cu(si), cu(hi)) if affine2 else None, w1d, stats=st1)
    torch.cuda.synchronize()
    assert torch.equal(a, a_ref), f"block output differs on {(a != a_ref).float().mean().item():.2e} of the elements"
    assert torch.equal(y1, y1_ref), f"conv1 output differs on {(y1 != y1_ref).float().mean().item():.2e} of the elements"
    assert_close(ops.stat_totals(st1).cpu(), ops.stat_totals(st1_ref).cpu(), 1e-6, what="conv1 statistics")
    # ... and against torch (f64) on the same bf16-rounded operands
    a2 = F.relu(y2.double() * s2.double()[None, :, None, None] + h2.double()[None, :, None, None]).bfloat16().double()
    t = F.conv2d(a2, w3.double()).bfloat16().double()
    ident = idt.double() * si.double()[None, :, None, None] + hi.double()[None, :, None, None] if affine2 else idt.double()
    aa = F.relu(t * s3.double()[None, :, None, None] + h3.double()[None, :, None, None] + ident)
    assert_close(a.float().cpu().permute(0, 3, 1, 2), aa, TOL, what="block output vs torch")
    assert_close(y1.float().cpu().permute(0, 3, 1, 2), F.conv2d(aa.bfloat16().double(), w1.double()), 2 * TOL, what="conv1 vs torch")


@pytest.mark.parametrize("geom", [(4, 16, 16, 64, 256), (2, 8, 8, 128, 512), (96, 32, 32, 64, 256), (40, 32, 32, 128, 512)], ids=lambda g: "x".join(map(str, g)))
def test_gram_statistics_give_the_convolutions_batchnorm_affine(gpu_lib, geom):
    """csrc/conv_gram.hip: train-mode BatchNorm (scale, shift, running statistics) of conv1x1(relu(bn(x)), w) from the Gram matrix of the
    convolution's INPUT (sum y_c = w_c . m, sum y_c^2 = w_c^T G w_c) against torch in f64 on the same bf16-rounded operand, and against the
    statistics-only launch it replaces (whose sums are of bf16-rounded outputs: agreement to that rounding's noise)"""
    from mhentropy_amd import ops, resnet
    B, H, W, Cb, C4 = geom
    g = torch.Generator().manual_seed(Cb + B)
    x = torch.randn(B, Cb, H, W, generator=g).bfloat16().float()
    w = (torch.randn(C4, Cb, 1, 1, generator=g) * (2.0 / Cb) ** 0.5).bfloat16().float()
    s2, h2 = torch.rand(Cb, generator=g) + 0.5, torch.randn(Cb, generator=g) * 0.3
    gamma, beta = torch.rand(C4, generator=g) + 0.5, torch.randn(C4, generator=g) * 0.2
    xd, wd = _nhwc(x), resnet.pack_conv_weight(w, torch.bfloat16).cuda()
    P = B * H * W
    rm, rv = torch.zeros(C4, device="cuda"), torch.ones(C4, device="cuda")
    nbt = torch.tensor(0, dtype=torch.int64, device="cuda")
    bufs = ops.gram_buffers(Cb, torch.device("cuda"))
    outs = []
    for rep in range(2):                                   # twice: a launch overwrites its partial slabs, nothing is carried over
        sc, sh, mi = ops.conv1x1_gram_bn(xd, s2.cuda(), h2.cuda(), wd, gamma.cuda(), beta.cuda(), rm, rv, bufs, num_batches_tracked=nbt,
                                         want_mean_invstd=True)
        outs.append((sc.clone(), sh.clone(), mi.clone()))
    assert int(nbt) == 2
    assert all(torch.equal(a, b) for a, b in zip(*outs))    # slab-order sums: bit-reproducible
    a = F.relu(x.double() * s2.double()[None, :, None, None] + h2.double()[None, :, None, None]).bfloat16().double()
    y = F.conv2d(a, w.double())
    mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    assert_close(mi[0].cpu(), mean, 2e-5, 1e-6, what="batch mean")
    assert_close(1.0 / mi[1].double().cpu() ** 2 - 1e-5, var, 1e-4, what="batch variance")
    sc_ref = gamma.double() / torch.sqrt(var + 1e-5)
    assert_close(sc.cpu(), sc_ref, 1e-4, what="scale")
    assert_close(sh.cpu(), beta.double() - mean * sc_ref, 1e-4, 1e-6, what="shift")
    rm_ref = 0.1 * mean * (1 + 0.9)
    assert_close(rm.cpu(), rm_ref, 1e-4, 1e-6, what="running_mean after two steps")
    rv_ref = 0.9 * (0.9 * 1.0 + 0.1 * var * P / (P - 1)) + 0.1 * var * P / (P - 1)
    assert_close(rv.cpu(), rv_ref, 1e-4, what="running_var after two steps")
    # the statistics-only launch (bf16-rounded outputs) agrees to the rounding noise of its sums
    st = ops.stat_unit(C4, "cuda")
    ops.conv1x1_stats(xd, wd, s2.cuda(), h2.cuda(), st)
    sc2, sh2 = ops.bn_finalize(st, gamma.cuda(), beta.cuda(), None, None, float(P))
    assert_close(sc.cpu(), sc2.cpu(), 2e-3, what="scale vs statistics-only launch")
    assert_close(sh.cpu(), sh2.cpu(), 2e-3, 1e-4, what="shift vs statistics-only launch")


@pytest.mark.parametrize("B", [1, 3])
def test_stem_with_the_max_pool_inside(gpu_lib, B):
    """csrc/stem_pool.hip: conv1 + batch statistics + 3x3 / stride-2 max pool of the RAW output in one kernel (per channel the window
    maximum where gamma >= 0, the minimum where gamma < 0) against the two-kernel path's tensors: the raw conv1 output of
    mhe_stem_conv7x7s2 pooled by torch (bit-identical values), its statistics, and maxpool(relu(bn(y))) = relu(scale * pooled + shift)"""
    from mhentropy_amd import ops, resnet
    g = torch.Generator().manual_seed(11 + B)
    x = torch.randn(B, 3, 256, 256, generator=g).cuda()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1)
    wd = resnet.pack_stem_weight(w, torch.bfloat16).cuda()
    gamma = torch.randn(64, generator=g)
    gamma[::5] = -gamma[::5].abs()                                 # negative scales: window minimum
    gamma[7] = 0.0
    S = ops.stat_shards()
    st_ref, st = ops.stat_unit(64, "cuda"), ops.stat_unit(64, "cuda")
    y = ops.stem_conv7x7s2(x, wd, torch.bfloat16, stats=st_ref)           # [B,128,128,64] raw
    assert ops.stem_pool_supported(B, 256, 256, torch.bfloat16) and not ops.stem_pool_supported(B, 128, 128, torch.bfloat16)
    p = ops.stem_conv7x7s2_pool(x, wd, gamma.cuda(), stats=st)
    torch.cuda.synchronize()
    yn = y.float().permute(0, 3, 1, 2)
    pmax = F.max_pool2d(yn, 3, 2, 1)
    pmin = -F.max_pool2d(-yn, 3, 2, 1)
    want = torch.where((gamma.cuda() >= 0)[None, :, None, None], pmax, pmin).permute(0, 2, 3, 1).contiguous().bfloat16()
    assert torch.equal(p, want), f"pooled raw output differs on {(p != want).float().mean().item():.2e} of the elements"
    assert_close(ops.stat_totals(st).cpu(), ops.stat_totals(st_ref).cpu(), 1e-6, what="conv1 batch statistics")
    # ... and the identity the consumers rely on
    sc, sh = gamma.cuda() * 0.7, torch.randn(64, generator=g).cuda()
    a_ref = F.max_pool2d(F.relu(yn * sc[None, :, None, None] + sh[None, :, None, None]), 3, 2, 1)
    a_new = F.relu(p.float().permute(0, 3, 1, 2) * sc[None, :, None, None] + sh[None, :, None, None])
    assert torch.equal(a_new, a_ref)


@pytest.mark.parametrize("geom", [(4, 16, 16, 64), (3, 32, 16, 128)], ids=lambda g: "x".join(map(str, g)))
def test_conv3_batchnorm_reverse_from_gram_statistics_against_autograd(gpu_lib, geom):
    """csrc/conv_fold.hip: reverse of z = BatchNorm_train(conv1x1(A, W)) for a given gradient g at z, computed WITHOUT y3 = conv(A, W) and
    without gy3: D = g^T A (one weight-gradient launch), the forward's Gram totals of A, sum_p g -> dgamma, dbeta, dW and the three
    operands (k2 W, W^T diag(k1) W, k0^T W) of the data-gradient launch; against torch autograd in f64 on the same bf16-rounded tensors.
    Tolerances: dW / dgamma / dbeta are f32 sums of bf16 products: 2e-3 of the tensor's scale; the data gradient passes through bf16 weights
    (k2 W and S rounded to bf16) and a bf16 residual: 2e-2."""
    from mhentropy_amd import ops, resnet
    B, H, W_, Cb = geom
    C4, P = 4 * Cb, B * H * W_
    gen = torch.Generator().manual_seed(Cb + B)
    A = (torch.rand(B, Cb, H, W_, generator=gen) + 0.0625).bfloat16().float()         # post-ReLU, all positive: the gate is open everywhere
    Wt = (torch.randn(C4, Cb, 1, 1, generator=gen) * (2.0 / Cb) ** 0.5).bfloat16().float()
    g = (torch.randn(B, C4, H, W_, generator=gen) * 0.1 + 0.03).bfloat16().float()
    gamma, beta = torch.rand(C4, generator=gen) + 0.5, torch.randn(C4, generator=gen) * 0.2
    # ---- autograd, f64
    Ad, Wd = A.double().requires_grad_(True), Wt.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    z = F.batch_norm(F.conv2d(Ad, Wd), None, None, gd, bd, True, 0.1, 1e-5)
    (z * g.double()).sum().backward()
    # ---- the fold
    Ax, gx = _nhwc(A), _nhwc(g)
    wd = resnet.pack_conv_weight(Wt, torch.bfloat16).cuda()
    ones, zeros = torch.ones(Cb, device="cuda"), torch.zeros(Cb, device="cuda")
    gbuf = (ops.gram_buffers(Cb, torch.device("cuda"))[0], ops.gram_workspace(Cb, torch.device("cuda")))
    rm, rv = torch.zeros(C4, device="cuda"), torch.ones(C4, device="cuda")
    _, _, mi = ops.conv1x1_gram_bn(Ax, ones, zeros, wd, gamma.cuda(), beta.cuda(), rm, rv, gbuf, want_mean_invstd=True)
    D = torch.zeros(C4, Cb, device="cuda")
    ops.conv_wgrad(Ax, gx, 1, 1, 1, 0, D)
    revf = torch.zeros(ops.stat_shards(), 2, C4, device="cuda")
    revf[3, 0] = g.sum((0, 2, 3)).cuda(); revf[:, 1] = 123.0                            # the second sum is not read
    rev = ops.stat_from_float(revf)
    dgamma, dbeta, dW = torch.zeros(C4, device="cuda"), torch.zeros(C4, device="cuda"), torch.zeros(C4, Cb, device="cuda")
    w_dg, S, c0 = torch.zeros(Cb, C4, device="cuda", dtype=torch.bfloat16), torch.zeros(Cb, Cb, device="cuda", dtype=torch.bfloat16), torch.zeros(Cb, device="cuda")
    ops.conv3_bn_fold(D, wd, gbuf[1], rev, gamma.cuda(), mi, P, dgamma, dbeta, dW, w_dg, S, c0, torch.zeros(2 * C4, device="cuda"))
    torch.cuda.synchronize()
    assert not D.any(), "the accumulator is left clean"
    assert_close(dbeta.cpu(), bd.grad, 2e-3, what="dbeta")
    assert_close(dgamma.cpu(), gd.grad, 2e-3, what="dgamma")
    assert_close(dW.cpu(), Wd.grad[:, :, 0, 0], 2e-3, what="dW")
    # data gradient: g (k2 W) + A S + c0, gated by [A > 0] (open everywhere here)
    t_res = ops.conv2d_nhwc(Ax, S, 1, 1, 1, 0)
    gA = ops.conv2d_nhwc(gx, w_dg, 1, 1, 1, 0, residual=t_res, mask=Ax, out_shift=c0)
    assert_close(gA.float().cpu().permute(0, 3, 1, 2), Ad.grad, 2e-2, what="data gradient")
    # ... as ONE launch on the K-concatenated operand [g | A] with weights [(k2 W)^T | S] (what the train step runs)
    D2 = torch.zeros(C4, Cb, device="cuda")
    ops.conv_wgrad(Ax, gx, 1, 1, 1, 0, D2)
    w_cat, c0b = torch.zeros(Cb, C4 + Cb, device="cuda", dtype=torch.bfloat16), torch.zeros(Cb, device="cuda")
    ops.conv3_bn_fold(D2, wd, gbuf[1], rev, gamma.cuda(), mi, P, torch.zeros(C4, device="cuda"), torch.zeros(C4, device="cuda"),
                      torch.zeros(C4, Cb, device="cuda"), w_cat, None, c0b, torch.zeros(2 * C4, device="cuda"))
    # (up to round 3 D came out of a split-K launch with f32 atomics and a second evaluation could differ in the last bit; slabs + a fixed-order reducer now)
    assert_close(w_cat[:, :C4].float().cpu(), w_dg.float().cpu(), 4e-3, what="(k2 W)^T in the concatenated weights")
    assert_close(w_cat[:, C4:].float().cpu(), S.float().cpu(), 4e-3, what="S in the concatenated weights")
    assert_close(c0b.cpu(), c0.cpu(), 1e-4, what="c0")
    gAc = ops.conv2d_nhwc(gx, w_cat, 1, 1, 1, 0, mask=Ax, out_shift=c0, xcat=Ax)
    assert_close(gAc.float().cpu().permute(0, 3, 1, 2), Ad.grad, 2e-2, what="data gradient, one concatenated launch")
    assert_close(gAc.float().cpu(), gA.float().cpu(), 1e-2, what="concatenated launch vs residual form")     # (the residual form rounds A S to bf16)
    # ... and the same launch with the consumer's BatchNorm-reverse sums (one bn triple) leaves the same tensor
    st = ops.stat_unit(Cb, "cuda")
    mi2 = torch.stack([torch.zeros(Cb), torch.ones(Cb)]).cuda()
    gA2 = ops.conv2d_nhwc(gx, w_dg, 1, 1, 1, 0, residual=t_res, mask=Ax, out_shift=c0, bn=[(Ax, mi2, st)])
    assert torch.equal(gA2, gA)
    # (sum_p of a train-mode BatchNorm's input gradient is zero: the sums are rounding noise - compared on the scale of sum |g|)
    err = (ops.stat_totals(st)[0] - gA.double().sum((0, 1, 2))).abs() / gA.double().abs().sum((0, 1, 2))
    assert float(err.max()) < 4e-3, float(err.max())


def test_relu_gate_as_bits_for_the_streaming_data_gradient(gpu_lib):
    """the block output's ReLU gate as bits (mhe_bottleneck_tail_bits_nhwc writes [a > 0], byte [pixel][channel / 8]) read by the streaming
    1x1 data-gradient kernel instead of the block-wide tensor (mhe_conv2d_masked_bits_nhwc): bits exactly [a > 0]; outputs and
    BatchNorm-reverse sums identical to the launch that reads the tensor; a consumer handed the gate tensor as its raw output gets sum g
    (and no second sum)."""
    from mhentropy_amd import ops, resnet
    B, H, W, Cb, N2 = 4, 32, 32, 64, 64
    C4 = 4 * Cb
    g = torch.Generator().manual_seed(5)
    y2 = torch.randn(B, Cb, H, W, generator=g).bfloat16().float()
    idt = torch.randn(B, C4, H, W, generator=g).bfloat16().float()
    w3 = (torch.randn(C4, Cb, 1, 1, generator=g) * (2.0 / Cb) ** 0.5).bfloat16().float()
    w1 = (torch.randn(N2, C4, 1, 1, generator=g) * (2.0 / C4) ** 0.5).bfloat16().float()
    s2, h2 = (torch.rand(Cb, generator=g) + 0.5).cuda(), (torch.randn(Cb, generator=g) * 0.3).cuda()
    s3, h3 = (torch.rand(C4, generator=g) + 0.5).cuda(), (torch.randn(C4, generator=g) * 0.3).cuda()
    w3d, w1d = resnet.pack_conv_weight(w3, torch.bfloat16).cuda(), resnet.pack_conv_weight(w1, torch.bfloat16).cuda()
    a, y1, bits = ops.bottleneck_tail(_nhwc(y2), (s2, h2), w3d, (s3, h3), _nhwc(idt), None, w1d, want_bits=True)
    a0, y10 = ops.bottleneck_tail(_nhwc(y2), (s2, h2), w3d, (s3, h3), _nhwc(idt), None, w1d)
    assert torch.equal(a, a0) and torch.equal(y1, y10)
    want = ((a.float() > 0).view(B, H, W, C4 // 8, 8).to(torch.int32) * (1 << torch.arange(8, device="cuda", dtype=torch.int32))).sum(-1).to(torch.uint8)
    assert torch.equal(bits, want)
    assert 0.2 < float((a.float() > 0).float().mean()) < 0.8
    # conv1's data gradient (64 -> 256 channels: the streaming kernel), gated by [a > 0], residual, two consumers' sums
    gy = torch.randn(B, H, W, N2, generator=g).bfloat16().cuda()
    res = torch.randn(B, H, W, C4, generator=g).bfloat16().cuda()
    ybn = torch.randn(B, H, W, C4, generator=g).bfloat16().cuda()
    wdg = resnet.pack_conv_weight(w1.permute(1, 0, 2, 3).contiguous(), torch.bfloat16).cuda()          # [C4][N2]
    assert ops.conv_tile_choice(B, H, W, N2, C4, 1, 1, 0, torch.bfloat16, 0) in (8, 1, 2, 7, 0)
    mi = torch.stack([torch.randn(C4, generator=g) * 0.1, torch.rand(C4, generator=g) + 0.5]).cuda().contiguous()
    S = ops.stat_shards()
    outs = []
    for mb in (None, bits):
        st0, st1 = ops.stat_unit(C4, "cuda"), ops.stat_unit(C4, "cuda")
        o = ops.conv2d_nhwc(gy, wdg, 1, 1, 1, 0, residual=res, mask=a, bn=[(ybn, mi, st0), (a, mi, st1)], mask_bits=mb)
        outs.append((o, ops.stat_totals(st0), ops.stat_totals(st1)))
    (o0, s00, s01), (o1, s10, s11) = outs
    assert torch.equal(o0, o1)
    assert_close(s10.cpu(), s00.cpu(), 1e-5, what="sums of the consumer with a raw output")
    assert_close(s11[0].cpu(), s01[0].cpu(), 1e-5, what="sum g of the consumer handed the gate tensor")
    assert_close(s11[0].cpu(), o1.float().sum((0, 1, 2)).cpu(), 2e-3, 1e-2, what="sum g against the stored gradient")


# B, H (= W), Cin, Cout: one tile per workgroup; an odd number of 64-channel chunks (the halo buffers change parity from tile to tile);
# two column tiles; more tiles than workgroups (the persistent loop: next tile staged during this one's last chunk) on both map sizes
HALO_GEOMS = [(4, 32, 128, 128), (3, 16, 192, 128), (5, 16, 256, 256), (80, 32, 64, 128), (136, 16, 192, 256)]


@pytest.mark.parametrize("form", ["plain", "bn-on-load", "data-gradient"])
@pytest.mark.parametrize("geom", HALO_GEOMS, ids=lambda g: "x".join(map(str, g)))
def test_3x3_kernel_with_the_input_tile_resident_in_lds(gpu_lib, geom, form):
    """conv_halo_kernel (csrc/conv_halo.hip: 3x3 / stride 1 / pad 1 on 32 x 32 and 16 x 16 maps; halo of 256 output pixels in LDS, weights
    streamed fragment-major): against conv2d on the bf16-rounded operands - producer BatchNorm + ReLU applied in f32 once per element and
    rounded, zero padding AFTER it, the normalised operand written out once and equal to the in-place pass's; statistics of the stored
    output; data-gradient form against the register-staged kernel"""
    from mhentropy_amd import ops, resnet
    B, H, Cin, Cout = geom
    W = H
    assert ops.conv3x3_halo_supported(B, H, W, Cin, Cout)
    g, x, w = _operands(B + Cin, B, H, W, Cin, Cout, 3)
    if B > 16:
        x = x.cuda(); w = w.cuda()
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    wd = resnet.pack_conv_weight(w, torch.bfloat16).cuda()
    wh = ops.conv3x3_halo_pack(wd)
    if form == "data-gradient":
        rnd = lambda *s: torch.randn(*s, generator=g).bfloat16().cuda()
        mask, res, by = rnd(B, H, W, Cout), rnd(B, H, W, Cout), rnd(B, H, W, Cout)
        mi = torch.stack([torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5]).cuda().contiguous()
        st = [ops.stat_unit(Cout, "cuda") for _ in range(2)]
        got = ops.conv3x3_halo(xd, wh, residual=res, mask=mask, bn=(by, mi, st[0]))
        want = ops.conv2d_nhwc(xd, wd, 3, 3, 1, 1, residual=res, mask=mask, bn=[(by, mi, st[1])], tile=2)
        assert_close(got.float().cpu(), want.float().cpu(), 8e-3, what="gated data gradient")
        a, b = ops.stat_totals(st[0]).cpu(), ops.stat_totals(st[1]).cpu()
        assert ((a - b).abs() <= 2e-2 * (b.abs().max(1, keepdim=True)[0] + 1.0)).all(), "BatchNorm-reverse sums"
        return
    kw, xin = {}, x
    a_out = None
    if form == "bn-on-load":
        sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
        a_out = torch.empty_like(xd)
        kw = dict(in_scale=sc.cuda(), in_shift=sh.cuda(), relu_in=True, a_out=a_out)
        xin = torch.relu(x * sc.to(x.device).view(1, -1, 1, 1) + sh.to(x.device).view(1, -1, 1, 1)).bfloat16().float()
    ref = F.conv2d(xin.double() if B <= 16 else xin, w.double() if B <= 16 else w, None, 1, 1)
    stats = ops.stat_unit(Cout, "cuda")
    y = ops.conv3x3_halo(xd, wh, stats=stats, **kw)
    assert_close(y.float().cpu().permute(0, 3, 1, 2), ref.cpu(), TOL, what="raw conv")
    n = ref.numel() / Cout
    stt = ops.stat_totals(stats).cpu()
    ys = y.double().cpu().permute(0, 3, 1, 2)
    assert_close(stt[0] / n, ys.mean((0, 2, 3)), 1e-5, 1e-5, what="batch mean")
    assert_close(stt[1] / n, (ys ** 2).mean((0, 2, 3)), 1e-5, what="batch E[x^2]")
    if a_out is not None:
        assert torch.equal(a_out, ops.bn_act(xd.clone(), kw["in_scale"], kw["in_shift"], relu=True)), "the normalised operand is the in-place pass's"
        assert_close(a_out.float().cpu(), xin.permute(0, 2, 3, 1).cpu(), 8e-3, what="normalised operand")      # (the kernels fuse the multiply-add)
    old = ops.conv2d_nhwc(xd, wd, 3, 3, 1, 1, in_scale=kw.get("in_scale"), in_shift=kw.get("in_shift"), relu_in=form == "bn-on-load", tile=2)
    assert_close(y.float().cpu(), old.float().cpu(), 8e-3, what="vs the tiled kernel")


def test_3x3_resident_tile_kernel_refuses_what_it_does_not_take(gpu_lib):
    from mhentropy_amd import ops, _lib
    assert not ops.conv3x3_halo_supported(4, 64, 64, 64, 64)        # 64-pixel rows: the row-streaming kernel's
    assert not ops.conv3x3_halo_supported(4, 8, 8, 512, 512)
    assert not ops.conv3x3_halo_supported(4, 24, 16, 128, 128)      # rows per tile do not divide the image
    assert not ops.conv3x3_halo_supported(4, 16, 16, 128, 64)
    x = torch.zeros(4, 8, 8, 512, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(512, 9 * 512, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(_lib.MheError):
        ops.conv3x3_halo(x, w)


@pytest.mark.parametrize("geom", [(4, 32, 128, 128), (5, 16, 256, 256), (80, 32, 128, 128)], ids=lambda g: "x".join(map(str, g)))
def test_3x3_resident_tile_data_gradient_with_batchnorm_reverse_on_its_load(gpu_lib, geom):
    """mhe_conv3x3_halo_dgrad_bn_nhwc: operand = k2 g + k1 y + k0 formed by the transfer waves (the arithmetic of the apply pass, so the
    tensor written out for the weight gradient is the pass's to the bit) - and then the same launch as on that tensor"""
    from mhentropy_amd import ops, resnet
    B, H, Cin, Cout = geom
    W = H
    g0, x, w = _operands(B + Cin + 1, B, H, W, Cin, Cout, 3)
    rnd = lambda *s: torch.randn(*s, generator=g0).bfloat16().cuda()
    g, y = x.permute(0, 2, 3, 1).contiguous().bfloat16().cuda(), rnd(B, H, W, Cin)
    wh = ops.conv3x3_halo_pack(resnet.pack_conv_weight(w, torch.bfloat16).cuda())
    mask, by = rnd(B, H, W, Cout), rnd(B, H, W, Cout)
    mi_in = torch.stack([torch.randn(Cin, generator=g0) * 0.1, torch.rand(Cin, generator=g0) + 0.5]).cuda().contiguous()
    mi_out = torch.stack([torch.randn(Cout, generator=g0) * 0.1, torch.rand(Cout, generator=g0) + 0.5]).cuda().contiguous()
    gamma = (torch.rand(Cin, generator=g0) + 0.5).cuda()
    S = ops.stat_shards()
    dg, db = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    sums = ops.stat_unit(Cin, "cuda")
    coef = ops.bn_backward(g, None, y, mi_in, gamma, sums, dg, db, coef_only=True)
    gy_ref = ops.bn_backward(g, None, y, mi_in, gamma, sums, dg, db, reduced=True)        # (the same sums: the same coefficients to the bit)
    st = [ops.stat_unit(Cout, "cuda") for _ in range(2)]
    gy = torch.empty_like(g)
    got = ops.conv3x3_halo_dgrad_bn(g, y, coef, wh, mask, gy_out=gy, bn=(by, mi_out, st[0]))
    want = ops.conv3x3_halo(gy_ref, wh, mask=mask, bn=(by, mi_out, st[1]))
    assert torch.equal(gy, gy_ref), (gy.float() - gy_ref.float()).abs().max().item()
    assert torch.equal(got, want)
    a, b = ops.stat_totals(st[0]).cpu(), ops.stat_totals(st[1]).cpu()
    assert ((a - b).abs() <= 1e-3 * (b.abs().max(1, keepdim=True)[0] + 1.0)).all(), "BatchNorm-reverse sums"
