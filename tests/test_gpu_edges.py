"""GPU: ragged / extreme shapes and the C-ABI's error behaviour (status codes + mhe_last_error, no
exceptions or faults across the boundary), plus size-independent properties at the bench's full size."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import assert_close
from mhentropy_amd import synth, mano_pack

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _blob():
    t = synth.mano_tables(0)
    return _dev(mano_pack.pack_tables(t["shapedirs"], t["posedirs"], t["v_template"], t["J_regressor"], t["weights"],
                                      t["hands_components"][:45], t["hands_mean"]))


def _flow(h, steps, bf16=False):
    from mhentropy_amd import ops
    sd = synth.flow_state(3, 45, 512, (h, h), steps)
    packs, b2, wc, bc = [], [], [], []
    for i in range(2 * steps):
        for net in ("s", "t"):
            p = f"{net}.{i}."
            f = ops.flow_pack_net_bf16 if bf16 else ops.flow_pack_net
            packs.append(f(sd[p + "l.0.weight"], sd[p + "l.1.weight"], sd[p + "l.2.weight"]))
            b2.append(sd[p + "l.2.bias"])
            for j in range(2):
                wc.append(sd[p + f"c.{j}.weight"]); bc.append(sd[p + f"c.{j}.bias"] + sd[p + f"l.{j}.bias"])
    ws = np.concatenate(packs)
    b2 = np.stack(b2)
    if bf16:
        ws, b2 = ws.view(np.int16), np.pad(b2, ((0, 0), (0, 19)))
    return sd, _dev(ws), _dev(b2), _dev(sd["mask"]), _dev(np.concatenate(wc)), _dev(np.concatenate(bc))


@pytest.mark.parametrize("B,N", [(1, 1), (1, 7), (5, 13), (3, 65), (2, 200)])
def test_flow_and_mano_ragged_row_counts(gpu_lib, B, N):
    """row counts that are not multiples of the 16/32/64-row wave and workgroup tiles"""
    from mhentropy_amd import ops
    from oracle import flows_ref, network_ref, mano_ref
    h, steps = 64, 2
    sd, ws, b2, mask, Wc, bc = _flow(h, steps)
    rng = np.random.default_rng(B * 100 + N)
    feat = rng.normal(0, 1, (B, 512)).astype(np.float32)
    z0 = rng.normal(0, 1, (N * B, 45)).astype(np.float32)
    cond = ops.linear(_dev(feat), Wc, bc).view(B, 4 * steps, 2, h)
    x, _, lq = ops.flow_couplings(_dev(z0), cond, ws, b2, mask, B, h, ops.FLOW_FORWARD)
    sdt = {k: torch.as_tensor(v) for k, v in sd.items()}
    with torch.no_grad():
        xr = flows_ref.forward_p(sdt, torch.as_tensor(z0), torch.as_tensor(feat).repeat(N, 1))
        lqr = flows_ref.log_prob(sdt, xr, torch.as_tensor(feat).repeat(N, 1))
    assert_close(x.cpu(), xr, 1e-4, what="x")
    assert_close(lq.cpu(), lqr, 1e-4, what="log q")
    # MANO loss pass on the same rows
    det = (rng.normal(0, 1, (B, 16)) * np.array([0.5] * 3 + [0.02] * 10 + [0.1] * 3)).astype(np.float32)
    _, yn = synth.batch(B, B, with_image=False)
    o = ops.mano_joints(x, _dev(det), _blob(), _dev(yn["crop_uv"]), _dev(yn["vis"]))
    tb = mano_ref.tables_from_numpy(synth.mano_tables(0))
    z = network_ref.combine_z(torch.as_tensor(det).repeat(N, 1), xr)
    with torch.no_grad():
        ref = network_ref.forward_log_p(tb, z, {k: torch.as_tensor(v) for k, v in yn.items()}, N)
    assert_close(o["log_p"].cpu(), ref["log_p"], 1e-4, what="log_p rows")
    verts = ops.mano_verts(o["z"], _blob())
    with torch.no_grad():
        dec = network_ref.decode(tb, z)
    assert_close(verts.cpu(), dec["verts"], 1e-4, what="verts")


def test_prior_terms_outside_their_support(gpu_lib):
    """the soft box / ball priors are zero on the golden vectors; drive them well outside the support"""
    from mhentropy_amd import ops
    from oracle import network_ref, mano_ref
    rng = np.random.default_rng(9)
    R = 6
    th45 = rng.normal(0, 2.5, (R, 45)).astype(np.float32)                 # beyond +-2
    det = np.zeros((R, 16), np.float32)
    det[:, :3] = rng.normal(0, 3.0, (R, 3))                               # |th3| beyond pi
    det[:, 3:13] = rng.normal(0, 0.08, (R, 10))                           # beyond +-0.03
    det[:, 13:] = rng.normal(0, 0.1, (R, 3))
    _, yn = synth.batch(1, R, with_image=False)
    o = ops.mano_joints(_dev(th45), _dev(det), _blob(), _dev(yn["crop_uv"]), _dev(yn["vis"]))
    tb = mano_ref.tables_from_numpy(synth.mano_tables(0))
    z = network_ref.combine_z(torch.as_tensor(det), torch.as_tensor(th45))
    with torch.no_grad():
        ref = network_ref.forward_log_p(tb, z, {k: torch.as_tensor(v) for k, v in yn.items()}, 1)
    for i, k in enumerate(("log_p_uv_giv_z", "log_p_th3", "log_p_th45", "log_p_bt")):
        assert float(ref[k].abs().max()) > 0
        assert_close(o["terms"][:, i].cpu(), ref[k], 1e-4, what=k)


def test_prior_terms_outside_their_support_match_reference_vectors(gpu_lib):
    """the same, against what the REFERENCE's own _forward_log_p returned (tests/golden/priors_outside.npz: th3 up to 3 pi,
    th45 up to +-5 and on the box faces, bt up to +-0.2; reference hand/network.py:155-165,429-435,612-667)"""
    from mhentropy_amd import ops
    from conftest import load_golden
    g = load_golden("priors_outside")
    o = ops.mano_joints(_dev(g["th45"]), _dev(g["det"]), _blob(), _dev(g["y_crop_uv"]), _dev(g["y_vis"]))
    assert_close(o["z"].cpu(), g["z"], 0.0, what="z assembly (pure copy)")
    for i, k in enumerate(("log_p_uv_giv_z", "log_p_th3", "log_p_th45", "log_p_bt")):
        assert_close(o["terms"][:, i].cpu(), g["terms_" + k], 1e-4, what=k)
    assert_close(o["log_p"].cpu(), g["terms_log_p"], 1e-4, what="log_p rows")


def test_vis_flags_other_than_one_are_masked(gpu_lib):
    """the reference masks with (vis == 1): 0 and 2 (H3.6M-style 'invisible') both drop the joint (network.py:255-257)"""
    from mhentropy_amd import ops
    B = 2
    rng = np.random.default_rng(2)
    th45 = rng.normal(0, 0.5, (B, 45)).astype(np.float32)
    det = (rng.normal(0, 0.1, (B, 16))).astype(np.float32)
    _, yn = synth.batch(3, B, with_image=False)
    vis0 = np.zeros((B, 21), np.float32)
    vis2 = np.full((B, 21), 2.0, np.float32)
    a = ops.mano_joints(_dev(th45), _dev(det), _blob(), _dev(yn["crop_uv"]), _dev(vis0))["terms"][:, 0].cpu()
    b = ops.mano_joints(_dev(th45), _dev(det), _blob(), _dev(yn["crop_uv"]), _dev(vis2))["terms"][:, 0].cpu()
    assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0


def test_c_abi_reports_errors_instead_of_faulting(gpu_lib):
    L = gpu_lib
    null = C.c_void_p(0)
    buf = torch.zeros(64, device="cuda")
    p = C.c_void_p(buf.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.mhe_linear_f32(null, p, null, p, 4, 8, 32, 0, s) != 0 and b"mhe_linear_f32" in L.mhe_last_error()
    assert L.mhe_linear_f32(p, p, null, p, 4, 8, 33, 0, s) != 0 and b"multiple of 32" in L.mhe_last_error()
    assert L.mhe_flow_couplings_f32(p, p, p, p, p, p, null, null, 10, 3, 45, 64, 4, 0, s) != 0       # R % B != 0
    assert b"multiple of B" in L.mhe_last_error()
    assert L.mhe_flow_couplings_f32(p, p, p, p, p, p, null, null, 12, 3, 45, 100, 4, 0, s) != 0      # hidden unsupported
    assert L.mhe_flow_couplings_f32(p, p, p, p, p, p, null, null, 12, 3, 45, 64, 4, 7, s) != 0       # bad direction
    assert L.mhe_mano_joints_f32(p, p, null, null, p, null, null, null, p, null, null, null, 4, 2, 0.03, 50.0, 0, 256.0, s) != 0
    assert b"crop_uv" in L.mhe_last_error()
    assert L.mhe_topk_gather_f32(p, p, p, p, 4, 2, 5, 45, s) != 0                                      # Q > N
    assert L.mhe_flow_packed_floats_per_net(45, 100) == 0
    torch.cuda.synchronize()                                                                           # nothing was launched, nothing faulted


def test_python_layer_rejects_wrong_inputs(gpu_lib):
    from mhentropy_amd import ops, _lib
    x = torch.zeros(4, 32, device="cuda")
    with pytest.raises(_lib.MheError):
        ops.linear(x.double(), torch.zeros(8, 32, device="cuda"))
    with pytest.raises(_lib.MheError):
        ops.linear(x.t(), torch.zeros(8, 4, device="cuda"))           # non-contiguous
    with pytest.raises(_lib.MheError):
        ops.mano_joints(torch.zeros(4, 44, device="cuda"), torch.zeros(2, 16, device="cuda"), _blob())


def test_full_size_properties(gpu_lib):
    """BASELINE size (R = 256 x 64 rows): properties that do not need the oracle at that size -
    inverse(forward(z)) == z, log q identical from both passes, rotations orthonormal through the
    bone-normalised joints being finite, ELBO reduce == torch mean."""
    from mhentropy_amd import ops
    B, N, h, steps = 256, 64, 512, 6
    sd, ws, b2, mask, Wc, bc = _flow(h, steps)
    g = torch.Generator(device="cuda").manual_seed(0)
    feat = torch.randn(B, 512, device="cuda", generator=g) * 0.5
    z0 = torch.randn(N * B, 45, device="cuda", generator=g)
    cond = ops.linear(feat, Wc, bc).view(B, 4 * steps, 2, h)
    x, sum_s, lq = ops.flow_couplings(z0, cond, ws, b2, mask, B, h, ops.FLOW_FORWARD)
    zb, sum_s2, lq2 = ops.flow_couplings(x, cond, ws, b2, mask, B, h, ops.FLOW_INVERSE)
    assert torch.isfinite(x).all()
    assert_close(zb.cpu(), z0.cpu(), 1e-4, what="inverse(forward(z)) at full size")
    assert_close(lq2.cpu(), lq.cpu(), 1e-4, what="log q from the inverse pass == from the sampling pass")
    det = torch.randn(B, 16, device="cuda", generator=g) * 0.05
    _, yn = synth.batch(0, B, with_image=False)
    o = ops.mano_joints(x, det, _blob(), _dev(yn["crop_uv"]), _dev(yn["vis"]))
    assert torch.isfinite(o["log_p"]).all() and torch.isfinite(o["xyz"]).all()
    xyz = o["xyz"].view(-1, 21, 3)
    assert_close((xyz[:, 11] - xyz[:, 12]).norm(dim=-1).cpu(), np.ones(N * B), 1e-5, what="unit reference bone")
    assert float(xyz[:, 12].abs().max()) < 1e-6                      # root at the origin
    q, hh, lp = ops.elbo_reduce(o["log_p"], lq, N, B)
    assert_close(q.cpu(), o["log_p"].view(N, B).mean(0).cpu(), 1e-5, what="mean over hypotheses")
    assert_close(lp.cpu(), (o["log_p"].view(N, B).mean(0) - lq.view(N, B).mean(0)).cpu(), 1e-5, what="ELBO")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_batch_statistics_of_large_magnitude_activations_stay_finite(gpu_lib, dt):
    """The fixed-point statistic accumulators (csrc/common.h, namespace fx) accept per-workgroup partial sums up to 2^38 (round 4: 2^32 -
    un-normalised 0..255 images or a diverging run could plant the NaN marker where the reference's BatchNorm, hand/network.py:54-61, goes
    on): activations of RMS ~3e3 give 256-pixel sums of y^2 of ~2^31 ... 2^35 - finite statistics equal to torch's in f64; beyond the
    range the finalize returns NaN (an explicit, detectable error), never a silently wrong number."""
    from mhentropy_amd import ops, resnet
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    B, H, W, Cin, Cout = 4, 32, 32, 64, 128
    x = (torch.randn(B, Cin, H, W, generator=g) * 400.0).to(dt).float()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5 * 6.0).to(dt).float()
    y_ref = F.conv2d(x.double(), w.double())
    assert float(y_ref.pow(2).mean().sqrt()) > 2.5e3
    xd = x.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    wd = resnet.pack_conv_weight(w, dt).cuda()
    st = ops.stat_unit(Cout, "cuda")
    y = ops.conv2d_nhwc(xd, wd, 1, 1, 1, 0, stats=st)
    tot = ops.stat_totals(st).cpu().double()
    assert torch.isfinite(tot).all()
    ys = y.double().cpu().permute(0, 3, 1, 2)            # the statistics are those of the output as stored
    n = B * H * W
    assert float(ys.pow(2).sum((0, 2, 3)).max()) > 2.0 ** 33          # the round-4 range (2^32 per partial) is crossed for whole channels
    assert_close(tot[0] / n, ys.mean((0, 2, 3)), 1e-5, 1e-3, what="batch mean at large magnitude")
    assert_close(tot[1] / n, ys.pow(2).mean((0, 2, 3)), 1e-5, what="batch E[y^2] at large magnitude")
    # ... and past the range: NaN, not garbage
    st2 = ops.stat_unit(Cout, "cuda")
    big = (xd.float() * 3.0e4).to(dt)
    ops.conv2d_nhwc(big, wd, 1, 1, 1, 0, stats=st2)
    ones, zeros = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    scale, _, mi = ops.bn_finalize(st2, ones, zeros, zeros.clone(), ones.clone(), n, want_mean_invstd=True)
    assert torch.isnan(mi[1]).all(), "sums of y^2 of ~1e19 per 256 pixels are beyond the accumulator's range: the finalize must return NaN"
    # (the in-range unit finalizes to torch's BatchNorm affine)
    scale1, shift1, mi1 = ops.bn_finalize(st, ones, zeros, zeros.clone(), ones.clone(), n, want_mean_invstd=True)
    var = ys.var((0, 2, 3), unbiased=False)
    assert_close(mi1[1].cpu(), 1.0 / torch.sqrt(var + 1e-5), 1e-4, what="invstd at large magnitude")
