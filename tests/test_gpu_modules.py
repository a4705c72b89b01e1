"""GPU parity at the reference's own module boundary: MHEnt.get_loss / sample,
MHEntLoss, ManoLayer.forward, RealNVP.log_prob/sample and the ResNet trunk,
against the golden vectors captured from the reference and against the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, assert_close
from mhentropy_amd import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-4          # BASELINE.json north_star: 1e-4 relative, fp32


def _t(d, dev="cuda"):
    return {k: torch.as_tensor(v).to(dev) for k, v in d.items()}


def _model_from_golden(g, backbone="resnet50"):
    from mhentropy_amd import harness
    seed, h, steps = int(g["seed"]), int(g["h"]), int(g["steps"])
    model = harness.build_mhent(backbone=backbone, h_dims=(h, h), num_steps=steps, tables=synth.mano_tables(0))
    sd = {"q_z_giv_i." + k: torch.as_tensor(v) for k, v in synth.flow_state(seed, 45, 512, (h, h), steps).items()}
    sd.update({k: torch.as_tensor(v) for k, v in synth.head_state(seed, 2048, 512, 16).items()})
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model = model.cuda()
    trunk = torch.as_tensor(g["trunk"]).cuda()
    model.feat_extractor.res.forward = lambda x: trunk        # the fixture pins everything after the trunk
    return model


@pytest.mark.parametrize("tag", ["small", "shipped"])
@pytest.mark.parametrize("fused", [True, False])
def test_get_loss_matches_reference_vectors(gpu_lib, tag, fused):
    g = load_golden(f"mhent_{tag}")
    model = _model_from_golden(g)
    model.fused_entropy = fused
    B = int(g["B"])
    y = _t({k[2:]: v for k, v in g.items() if k.startswith("y_")})
    x = torch.zeros(B, 3, 8, 8, device="cuda")
    out = model.get_loss(x, y, mods=["uv"], noise=torch.as_tensor(g["z0_loss"]).cuda())
    assert set(out) == {"th_norm", "bt_norm", "log_p", "q_log_p_z_giv_y", "h_q_z_giv_i"}
    for k in out:
        assert_close(out[k].cpu(), g["loss_" + k], RTOL, what=k)


@pytest.mark.parametrize("tag,N", [("small", 4), ("shipped", 16)])
def test_other_hypothesis_counts(gpu_lib, tag, N):
    g = load_golden(f"mhent_{tag}")
    model = _model_from_golden(g)
    B = int(g["B"])
    feat = torch.as_tensor(g["feat"]).cuda()
    z0 = torch.as_tensor(g[f"z0_N{N}"]).cuda()
    th45, lq = model.q_z_giv_i.sample_with_log_prob(z0, feat)
    assert_close(th45.cpu(), g[f"z_N{N}"][:, 3:48], RTOL, what="th45")
    assert_close(lq.cpu(), g[f"logq_N{N}"], RTOL, what="log q (fused)")
    assert_close(model.q_z_giv_i.log_prob(th45, logvar=feat).cpu(), g[f"logq_N{N}"], RTOL, what="log q (inverse pass)")
    from mhentropy_amd import ops
    o = ops.mano_joints(th45, model._det(feat), model.mano_dec.table_blob(), torch.as_tensor(g["y_crop_uv"]).cuda(),
                        torch.as_tensor(g["y_vis"]).cuda(), want=("log_p", "z"))
    assert_close(o["z"].cpu(), g[f"z_N{N}"], RTOL, what="z")
    assert_close(o["log_p"].cpu(), g[f"logp_rows_N{N}"], RTOL, what="log_p rows")


@pytest.mark.parametrize("tag", ["small", "shipped"])
def test_sample_and_criterion_match_reference_vectors(gpu_lib, tag):
    from mhentropy_amd.criteria import MHEntLoss
    g = load_golden(f"mhent_{tag}")
    model = _model_from_golden(g)
    B = int(g["B"])
    y = _t({k[2:]: v for k, v in g.items() if k.startswith("y_")})
    x = torch.zeros(B, 3, 8, 8, device="cuda")
    s = model.sample(x, N=[4, 4], temp=0.8, mods={"uv", "xyz", "verts"}, y=y,
                     noise=torch.as_tensor(g["z0_sample"]).cuda() / 0.8)
    for k in ("th_bt", "logs_t", "verts", "xyz", "uv"):
        assert tuple(s[k].shape) == g["sample_" + k].shape, k
        assert_close(s[k].cpu(), g["sample_" + k], RTOL, what="sample." + k)
    assert torch.equal(s["faces"].cpu(), torch.as_tensor(synth.mano_tables(0)["faces"]))
    out = {"log_p": torch.as_tensor(g["loss_log_p"]).cuda(), "xyz": s["xyz"], "uv": s["uv"], "verts": s["verts"]}
    total, losses, metrics = MHEntLoss()(out, y)
    assert_close(total.cpu(), g["criterion_total"], RTOL, what="total")
    assert set(losses) == {"neg_log_p"}
    assert len(metrics) == 14
    for k, v in metrics.items():
        assert_close(v.cpu(), g["metric_" + k], 5e-4, what=k)      # metrics amplify the 1e-4 of xyz/uv through std/min


@pytest.mark.parametrize("tag", ["small", "shipped"])
def test_sample_topk_matches_reference_vectors(gpu_lib, tag):
    """MHEnt.sample(N=[6,3]): the 3 most likely of 6 hypotheses per image (hand/network.py:866-871)"""
    g = load_golden(f"mhent_{tag}")
    model = _model_from_golden(g)
    B = int(g["B"])
    y = _t({k[2:]: v for k, v in g.items() if k.startswith("y_")})
    s = model.sample(torch.zeros(B, 3, 8, 8, device="cuda"), N=[6, 3], temp=0.8, mods={"uv", "xyz", "verts"}, y=y,
                     noise=torch.as_tensor(g["z0_topk"]).cuda() / 0.8)
    for k in ("th_bt", "logs_t", "verts", "xyz", "uv"):
        assert tuple(s[k].shape) == g["topk_" + k].shape, k
        assert_close(s[k].cpu(), g["topk_" + k], RTOL, what="topk." + k)


def test_mano_layer_forward_matches_reference_vectors(gpu_lib):
    from mhentropy_amd.ManoLayer import ManoLayer
    g = load_golden("mano")
    layer = ManoLayer(skeidx="RHD", flat_hand_mean=False, ncomps=45, use_pca=True, tables=synth.mano_tables(0)).cuda()
    out = layer(beta=torch.as_tensor(g["beta"]).cuda(), theta=torch.as_tensor(g["theta"]).cuda())
    for k in ("mesh", "mano_joints", "joints"):
        assert_close(out[k].cpu(), g[k], RTOL, what=k)


@pytest.mark.parametrize("arch,B,S", [("resnet18", 4, 64), ("resnet50", 4, 64)])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("mode", ["pass+fused-tail", "load+fused-tail", "pass", "load"])
def test_resnet_trunk_matches_oracle(gpu_lib, arch, B, S, training, mode):
    from mhentropy_amd import resnet
    from oracle import resnet_ref
    sdn = synth.resnet_state(3, arch)
    x, _ = synth.batch(3, B, image_size=S)
    trunk = resnet.ResNetTrunk(arch)
    trunk.bn_apply = trunk.bn_apply_1x1 = mode.split("+")[0]
    trunk.fuse_tail = mode.endswith("fused-tail")
    trunk.load_state_dict({k: torch.as_tensor(v) for k, v in sdn.items()})
    trunk = trunk.cuda().train(training)
    f = trunk(torch.as_tensor(x).cuda())
    stats = {}
    with torch.no_grad():
        ref = resnet_ref.forward({k: torch.as_tensor(v) for k, v in sdn.items()}, torch.as_tensor(x), arch, training, stats)
    assert_close(f.cpu(), ref, RTOL, what="pooled feature")
    if training:      # running statistics updated like torch's BatchNorm
        mean, var, rm, rv = stats["layer1.0.bn1"]
        assert_close(trunk.layer1[0].bn1.running_mean.cpu(), rm, RTOL, 1e-6, what="running_mean")
        assert_close(trunk.layer1[0].bn1.running_var.cpu(), rv, RTOL, what="running_var")
        assert int(trunk.bn1.num_batches_tracked) == 1


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
@pytest.mark.parametrize("bn_apply", ["pass", "load"])
def test_resnet_trunk_bf16_storage_mode(gpu_lib, arch, bn_apply, monkeypatch):
    """bf16 is a performance mode, checked against the oracle that rounds at the same storage points
    (f32 accumulate everywhere).  Two roundings of values that differ by 1e-6 disagree by a full bf16 ulp
    (4e-3) on a few elements, and after a handful of layers on all of them, so agreement is asserted
    tightly on the first residual block (where the implementations are still in lock-step), and on the
    pooled feature within the band by which bf16 storage itself moves the f32 result."""
    from mhentropy_amd import resnet, ops
    from oracle import resnet_ref
    B, S = 8, 128        # >= 128 samples per channel in every BatchNorm
    sdn = synth.resnet_state(4, arch)
    sd = {k: torch.as_tensor(v) for k, v in sdn.items()}
    x, _ = synth.batch(4, B, image_size=S)
    trunk = resnet.ResNetTrunk(arch, compute_dtype=torch.bfloat16)
    trunk.bn_apply = trunk.bn_apply_1x1 = bn_apply
    trunk.load_state_dict(sd)
    trunk = trunk.cuda().train()
    taps, orig = [], ops.bn_act

    trunk.fuse_tail = False                      # the spy below needs every block tail materialised by bn_act

    def spy(x_, scale, shift, res=None, *a, **k):
        y = orig(x_, scale, shift, res, *a, **k)
        if res is not None:                      # the tail of a residual block
            taps.append(y.float().cpu().permute(0, 3, 1, 2).clone())
        return y
    monkeypatch.setattr(ops, "bn_act", spy)
    f = trunk(torch.as_tensor(x).cuda()).cpu()
    ref_taps = {}
    with torch.no_grad():
        ref = resnet_ref.forward_bf16_storage(sd, torch.as_tensor(x), arch, True, taps=ref_taps)
        ref32 = resnet_ref.forward(sd, torch.as_tensor(x), arch, True)
    first = ref_taps["layer1.0"]
    e0 = (taps[0] - first).abs().mean().item() / first.abs().mean().item()
    err = (f - ref).abs().mean().item() / ref.abs().mean().item()
    err32 = (ref32 - ref).abs().mean().item() / ref.abs().mean().item()
    print(f"bf16 trunk {arch}/{bn_apply}: first block mean-rel {e0:.2e}; pooled feature mean-rel {err:.2e} "
          f"(bf16-storage oracle vs f32 oracle: {err32:.2e})")
    assert e0 < 1e-3, e0
    assert err < max(1e-2, 1.5 * err32), (err, err32)


def test_row_streamed_3x3_bn_on_load_equals_in_place_pass(gpu_lib):
    """At C2's geometry layer1's 3x3 convolutions run on the row-streaming kernel, which takes the producer's BatchNorm on its row load
    (resnet.bn_apply_3x3 = "auto").  The in-place pass rounds relu(y * scale + shift) to bf16 when it stores, the kernel rounds the same
    f32 value on its way into LDS: with the running statistics (eval) both trunks see the same affine maps and must agree to the bit;
    with batch statistics too (the sums are order-independent fixed point since round 4: they were float atomics whose order changed from
    launch to launch, and the two could only agree as closely as two runs of one policy did)."""
    from mhentropy_amd import resnet, ops
    B, S = 128, 256
    assert ops.conv_tile_choice(B, S // 4, S // 4, 64, 64, 3, 1, 1, torch.bfloat16, 1) == 9
    sd = {k: torch.as_tensor(v) for k, v in synth.resnet_state(5, "resnet50").items()}
    x = torch.as_tensor(synth.batch(5, 8, image_size=S)[0]).cuda().repeat(B // 8, 1, 1, 1)
    x = x + 0.01 * torch.randn(x.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    feats = {}
    for training in (False, True):
        for policy in ("auto", "pass", "pass again", "auto + resident tile"):
            trunk = resnet.ResNetTrunk("resnet50", compute_dtype=torch.bfloat16)
            trunk.bn_apply_3x3 = policy.split()[0]
            # (layer2 / layer3's 3x3 kernel with the input tile resident in LDS, csrc/conv_halo.hip, normalises on its load as well but sums
            # in another order than the im2col kernels: it is held to a tolerance below, the row-streaming kernel to the bit;
            # the stride-2 3x3 launches take the BatchNorm on their load in both "auto" forms and in place under "pass": same kernel, same sums)
            trunk.conv_halo = policy.endswith("resident tile")
            trunk.load_state_dict(sd)
            trunk = trunk.cuda().train(training)
            feats[policy] = trunk(x)
        assert torch.isfinite(feats["auto"]).all() and torch.isfinite(feats["auto + resident tile"]).all()
        if not training:
            assert torch.equal(feats["auto"], feats["pass"]), (feats["auto"] - feats["pass"]).abs().max().item()
            scale = feats["pass"].abs().mean().item()
            err = (feats["auto + resident tile"] - feats["pass"]).abs().mean().item() / scale
            print(f"resident-tile 3x3 kernel vs in-place pass + im2col kernel, running statistics: pooled feature mean-rel {err:.2e}")
            assert err < 5e-3, err
        else:
            scale = feats["pass"].abs().mean().item()
            err = (feats["auto"] - feats["pass"]).abs().mean().item() / scale
            rerun = (feats["pass again"] - feats["pass"]).abs().mean().item() / scale
            print(f"3x3 BatchNorm on load vs in place, batch statistics: pooled feature mean-rel {err:.2e} (the in-place policy run twice: {rerun:.2e})")
            # (round 4: the statistics are order-independent fixed point - two runs of one policy, and the row-streaming kernel against
            # the in-place pass it replaces, agree to the BIT)
            assert rerun == 0.0 and err == 0.0, (err, rerun)
            err_h = (feats["auto + resident tile"] - feats["pass"]).abs().mean().item() / scale
            print(f"   ... with the resident-tile kernel: {err_h:.2e}")
            # another summation order inside the product (measured 1.1e-2: a bf16 rounding that flips, then 53 train-mode BatchNorms)
            assert err_h < 2.5e-2, err_h


def test_full_path_end_to_end_vs_oracle(gpu_lib):
    """config C0 (BASELINE.json configs[0]): ResNet-18, 2-block small flow, K=4, B=2, 256x256."""
    from mhentropy_amd import harness
    from oracle import network_ref, mano_ref
    B, N, h, steps = 2, 4, 64, 2
    sdn = {"q_z_giv_i." + k: v for k, v in synth.flow_state(31, 45, 512, (h, h), steps).items()}
    sdn.update(synth.head_state(31, 512, 512, 16))
    sdn.update({"feat_extractor.res." + k: v for k, v in synth.resnet_state(31, "resnet18").items()})
    x, yn = synth.batch(31, B, image_size=256)
    z0 = synth.noise(31, N * B)
    model = harness.build_mhent(backbone="resnet18", h_dims=(h, h), num_steps=steps, tables=synth.mano_tables(0))
    missing, unexpected = model.load_state_dict({k: torch.as_tensor(v) for k, v in sdn.items()}, strict=False)
    assert not unexpected and all(k.startswith("mano_dec.") for k in missing)
    model = model.cuda().train()
    out = model.get_loss(torch.as_tensor(x).cuda(), _t(yn), mods=["uv"], N=N, noise=torch.as_tensor(z0).cuda())
    sd = {k: torch.as_tensor(v) for k, v in sdn.items()}
    tb = mano_ref.tables_from_numpy(synth.mano_tables(0))
    with torch.no_grad():
        ref = network_ref.get_loss(sd, tb, torch.as_tensor(x), _t(yn, "cpu"), torch.as_tensor(z0), N, "resnet18", True)
    for k in ("th_norm", "bt_norm", "q_log_p_z_giv_y", "h_q_z_giv_i", "log_p"):
        assert_close(out[k].cpu(), ref[k], 2e-4, what=k)       # 18 BN layers of round-off in front of the 1e-4 path


def test_resnet50_trunk_256x256_f32_matches_oracle(gpu_lib):
    """the encoder at the reference's input size (hand/network.py:54-61,110: torchvision resnet50 on 256x256 crops) in f32, train-mode
    BatchNorm, against the CPU restatement; B = 6 keeps the oracle at a few seconds and makes the launcher pick the large tiles on
    layer1/2 (M = 6*64*64 = 24,576 pixels)."""
    from mhentropy_amd import resnet
    from oracle import resnet_ref
    B = 6
    sdn = synth.resnet_state(17, "resnet50")
    sd = {k: torch.as_tensor(v) for k, v in sdn.items()}
    x, _ = synth.batch(17, B, image_size=256)
    trunk = resnet.ResNetTrunk("resnet50")
    trunk.load_state_dict(sd)
    trunk = trunk.cuda().train()
    f = trunk(torch.as_tensor(x).cuda())
    stats = {}
    with torch.no_grad():
        ref = resnet_ref.forward(sd, torch.as_tensor(x), "resnet50", True, stats)
    assert f.shape == (B, 2048)
    assert_close(f.cpu(), ref, RTOL, what="pooled feature, ResNet-50 @ 256x256")
    for name, mod in (("layer1.0.bn1", trunk.layer1[0].bn1), ("layer4.2.bn3", trunk.layer4[2].bn3)):
        mean, var, rm, rv = stats[name]
        assert_close(mod.running_mean.cpu(), rm, RTOL, 1e-6, what=name + ".running_mean")
        assert_close(mod.running_var.cpu(), rv, RTOL, what=name + ".running_var")


def test_c1_full_size_f32_vs_oracle_and_properties(gpu_lib):
    """config C1 (BASELINE.json configs[1]): ResNet-50 + the shipped 12-coupling h=512 RealNVP, B = 64 images x K = 16 hypotheses, f32,
    forward + loss at 256x256 - the whole workload against the CPU restatement (about 10 s of oracle time), plus two properties
    that hold at any size: (i) the per-image terms do not depend on the order of an image's hypotheses (noise rows permuted within
    each image: K-means agree to summation-order round-off), (ii) the K hypotheses of an image see the same conditioning: feeding the
    same noise row K times gives K identical rows (h_q_z_giv_i = that row's -log q, q_log_p = its own value)."""
    from mhentropy_amd import harness
    from oracle import network_ref, mano_ref
    B, N = 64, 16
    sdn = {"q_z_giv_i." + k: v for k, v in synth.flow_state(41, 45, 512, (512, 512), 6).items()}
    sdn.update(synth.head_state(41, 2048, 512, 16))
    sdn.update({"feat_extractor.res." + k: v for k, v in synth.resnet_state(41, "resnet50").items()})
    x, yn = synth.batch(41, B, image_size=256)
    z0 = synth.noise(41, N * B)
    model = harness.build_mhent(backbone="resnet50", h_dims=(512, 512), num_steps=6, tables=synth.mano_tables(0))
    missing, unexpected = model.load_state_dict({k: torch.as_tensor(v) for k, v in sdn.items()}, strict=False)
    assert not unexpected and all(k.startswith("mano_dec.") for k in missing)
    model = model.cuda().train()
    xg, yg, zg = torch.as_tensor(x).cuda(), _t(yn), torch.as_tensor(z0).cuda()
    keys = ("th_norm", "bt_norm", "q_log_p_z_giv_y", "h_q_z_giv_i", "log_p")
    out = {k: v.clone() for k, v in model.get_loss(xg, yg, mods=["uv"], N=N, noise=zg).items() if k in keys}
    sd = {k: torch.as_tensor(v) for k, v in sdn.items()}
    tb = mano_ref.tables_from_numpy(synth.mano_tables(0))
    with torch.no_grad():
        ref = network_ref.get_loss(sd, tb, torch.as_tensor(x), _t(yn, "cpu"), torch.as_tensor(z0), N, "resnet50", True)
        # the arbiter: the same oracle in float64.  north_star's bar is 1e-4 of the reference's fp32 CPU path; that path is itself
        # 53 train-mode BatchNorm layers of fp32 round-off away from the exact arithmetic, so the claim made (and asserted) here is
        # |HIP - f64| <= max(1e-4 scale, 2 |f32 oracle - f64|) per entry, with the measured figures printed
        D = torch.float64
        sd64 = {k: (v.to(D) if v.is_floating_point() else v) for k, v in sd.items()}
        y64 = {k: (v.to(D) if v.is_floating_point() else v) for k, v in _t(yn, "cpu").items()}
        ref64 = network_ref.get_loss(sd64, mano_ref.tables_from_numpy(synth.mano_tables(0), dtype=D), torch.as_tensor(x).to(D), y64,
                                     torch.as_tensor(z0).to(D), N, "resnet50", True)
    for k in keys:
        assert out[k].shape == ref[k].shape
        a, r32, r64 = out[k].cpu().double(), ref[k].double(), ref64[k].double()
        scale = float(r64.abs().max())
        e_hip, e_f32, e_pair = float((a - r64).abs().max()) / scale, float((r32 - r64).abs().max()) / scale, float((a - r32).abs().max()) / scale
        print(f"C1 f32 {k}: |HIP-f64| {e_hip:.2e}  |f32 oracle-f64| {e_f32:.2e}  |HIP-f32 oracle| {e_pair:.2e}  (max-norm, relative to max|f64|)")
        assert e_hip <= max(1e-4, 2.0 * e_f32), (k, e_hip, e_f32)
        # north_star's bar itself: 1e-4 of the reference's fp32 CPU path (measured on MI355X, round 5: th_norm 4.4e-6, bt_norm 1.1e-5, q_log_p
        # 2.2e-6, h 1.3e-6, log_p 2.2e-6 - rounds 3-4 asserted 3e-4 without having measured)
        assert_close(out[k].cpu(), ref[k], 1e-4, what="C1 " + k)
    # (i) hypothesis order within an image is irrelevant (rows are sample-major: r = n*B + b)
    perm = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b)) for b in range(B)], 1).cuda()      # (N, B)
    zp = zg.view(N, B, -1).gather(0, perm[:, :, None].expand(-1, -1, zg.shape[-1])).reshape(N * B, -1).contiguous()
    outp = model.get_loss(xg, yg, mods=["uv"], N=N, noise=zp)
    for k in ("q_log_p_z_giv_y", "h_q_z_giv_i", "log_p"):
        assert_close(outp[k].cpu(), out[k].cpu(), 2e-5, what="permuted hypotheses " + k)
    # (ii) K copies of one noise row: per-image means equal the single-hypothesis values
    z1 = zg.view(N, B, -1)[:1].expand(N, -1, -1).reshape(N * B, -1).contiguous()
    o1 = {k: v.clone() for k, v in model.get_loss(xg, yg, mods=["uv"], N=N, noise=z1).items() if k in keys}
    oK = model.get_loss(xg, yg, mods=["uv"], N=1, noise=zg[:B].contiguous())
    for k in ("q_log_p_z_giv_y", "h_q_z_giv_i", "log_p"):
        assert_close(o1[k].cpu(), oK[k].cpu(), 2e-5, what="repeated hypothesis " + k)


_C1 = {}


def _c1_case():
    """config C1's networks, inputs and the f32 oracle's loss dict (about 10 s of CPU time), shared by the tests that need them"""
    if not _C1:
        from oracle import network_ref, mano_ref
        B, N = 64, 16
        sdn = {"q_z_giv_i." + k: v for k, v in synth.flow_state(41, 45, 512, (512, 512), 6).items()}
        sdn.update(synth.head_state(41, 2048, 512, 16))
        sdn.update({"feat_extractor.res." + k: v for k, v in synth.resnet_state(41, "resnet50").items()})
        x, yn = synth.batch(41, B, image_size=256)
        z0 = synth.noise(41, N * B)
        sd = {k: torch.as_tensor(v) for k, v in sdn.items()}
        tb = mano_ref.tables_from_numpy(synth.mano_tables(0))
        with torch.no_grad():
            ref = network_ref.get_loss(sd, tb, torch.as_tensor(x), _t(yn, "cpu"), torch.as_tensor(z0), N, "resnet50", True)
        _C1.update(B=B, N=N, sdn=sdn, x=x, yn=yn, z0=z0, ref=ref)
    return _C1


def test_c1_shape_bf16_mode_deviation_from_the_f32_oracle(gpu_lib):
    """The timed product path is the bf16 mode (bf16 operands / storage, f32 accumulation).  Its pieces are checked against oracles that
    round at the same points; THIS test states how far the whole `get_loss` dict lands from the f32 reference value on config C1's
    shape (ResNet-50, shipped flow, B = 64, K = 16, 256x256): measured deviation printed, bounds asserted.  Not a 1e-4 claim - the
    number the headline rests on.  Per-image entries are compared by mean |a - b| / mean |b| (and the batch mean of log_p, the loss
    value itself); th_norm / bt_norm per hypothesis row."""
    from mhentropy_amd import harness
    c = _c1_case()
    model = harness.build_mhent(backbone="resnet50", h_dims=(512, 512), num_steps=6, tables=synth.mano_tables(0), compute_dtype=torch.bfloat16)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in c["sdn"].items()}, strict=False)
    model = model.cuda().train()
    out = model.get_loss(torch.as_tensor(c["x"]).cuda(), _t(c["yn"]), mods=["uv"], N=c["N"], noise=torch.as_tensor(c["z0"]).cuda())
    dev = {}
    for k in ("th_norm", "bt_norm", "q_log_p_z_giv_y", "h_q_z_giv_i", "log_p"):
        a, b = out[k].cpu().double(), c["ref"][k].double()
        assert a.shape == b.shape and torch.isfinite(a).all(), k
        dev[k] = ((a - b).abs().mean() / b.abs().mean()).item()
    loss_rel = abs(float(out["log_p"].mean()) - float(c["ref"]["log_p"].mean())) / abs(float(c["ref"]["log_p"].mean()))
    print("C1 bf16 mode vs f32 oracle, mean|a-b|/mean|b|: " + ", ".join(f"{k} {v:.2e}" for k, v in dev.items()) + f"; loss value {loss_rel:.2e}")
    # bounds = twice what was measured on MI355X (th_norm 9.0e-3, bt_norm 2.3e-2, q_log_p 4.4e-3, h 2.5e-3, log_p 4.7e-3,
    # loss 3e-4) - the path is deterministic since round 4 (fixed-point statistics), so the figures no longer move from run to run;
    # the entropy term only sees the flow (bf16 products), the likelihood term is the sensitive one (Laplace scale b = 0.03 on 42
    # re-projected coordinates)
    assert dev["h_q_z_giv_i"] < 5e-3 and dev["th_norm"] < 2e-2 and dev["bt_norm"] < 5e-2, dev
    assert dev["q_log_p_z_giv_y"] < 1e-2 and dev["log_p"] < 1e-2 and loss_rel < 1e-3, (dev, loss_rel)


def test_c2_full_size_vs_oracle(gpu_lib):
    """config C2 WHOLE - the workload the headline is timed on (BASELINE.json configs[2]: ResNet-50, shipped 12-coupling h = 512 RealNVP,
    B = 256 images x K = 64 hypotheses, 256x256) - against the CPU restatement on the same seeded inputs (one oracle pass: ~15 s on the
    box's 16 cores, the pass bench.py times as cpu_baseline):
      f32 mode   every `get_loss` entry at north_star's 1e-4 (measured figures printed),
      bf16 mode  (the timed product path) the deviation from the f32 reference value, printed and bounded at the C1 figures."""
    from mhentropy_amd import harness
    from oracle import network_ref, mano_ref
    B, N = 256, 64
    sdn = {"q_z_giv_i." + k: v for k, v in synth.flow_state(43, 45, 512, (512, 512), 6).items()}
    sdn.update(synth.head_state(43, 2048, 512, 16))
    sdn.update({"feat_extractor.res." + k: v for k, v in synth.resnet_state(43, "resnet50").items()})
    x, yn = synth.batch(43, B, image_size=256)
    z0 = synth.noise(43, N * B)
    sd = {k: torch.as_tensor(v) for k, v in sdn.items()}
    tb = mano_ref.tables_from_numpy(synth.mano_tables(0))
    with torch.no_grad():
        ref = network_ref.get_loss(sd, tb, torch.as_tensor(x), _t(yn, "cpu"), torch.as_tensor(z0), N, "resnet50", True)
    keys = ("th_norm", "bt_norm", "q_log_p_z_giv_y", "h_q_z_giv_i", "log_p")
    xg, yg, zg = torch.as_tensor(x).cuda(), _t(yn), torch.as_tensor(z0).cuda()
    for dt in (torch.float32, torch.bfloat16):
        model = harness.build_mhent(backbone="resnet50", h_dims=(512, 512), num_steps=6, tables=synth.mano_tables(0), compute_dtype=dt)
        model.load_state_dict({k: torch.as_tensor(v) for k, v in sdn.items()}, strict=False)
        model = model.cuda().train()
        out = {k: v.clone() for k, v in model.get_loss(xg, yg, mods=["uv"], N=N, noise=zg).items() if k in keys}
        del model
        torch.cuda.empty_cache()
        if dt == torch.float32:
            for k in keys:
                assert out[k].shape == ref[k].shape
                e = float((out[k].cpu().double() - ref[k].double()).abs().max() / ref[k].double().abs().max())
                print(f"C2 f32 {k}: |HIP - f32 oracle| {e:.2e} (max-norm, relative to max|oracle|)")
                assert_close(out[k].cpu(), ref[k], 1e-4, what="C2 f32 " + k)          # north_star's 1e-4 (3e-4 up to round 4)
            continue
        dev = {}
        for k in keys:
            a, b = out[k].cpu().double(), ref[k].double()
            assert a.shape == b.shape and torch.isfinite(a).all(), k
            dev[k] = ((a - b).abs().mean() / b.abs().mean()).item()
        loss_rel = abs(float(out["log_p"].mean()) - float(ref["log_p"].mean())) / abs(float(ref["log_p"].mean()))
        print("C2 bf16 mode vs f32 oracle, mean|a-b|/mean|b|: " + ", ".join(f"{k} {v:.2e}" for k, v in dev.items()) + f"; loss value {loss_rel:.2e}")
        assert dev["h_q_z_giv_i"] < 5e-3 and dev["th_norm"] < 2e-2 and dev["bt_norm"] < 5e-2, dev
        assert dev["q_log_p_z_giv_y"] < 1e-2 and dev["log_p"] < 1e-2 and loss_rel < 1e-3, (dev, loss_rel)


def test_c2_size_bf16_graph_replay_equals_eager(gpu_lib):
    """config C2's forward + loss (B = 256, K = 64, bf16) is DETERMINISTIC like the reference's CPU `get_loss` (hand/network.py:838-844):
    two eager runs and the HIP-graph replay the bench times return the same bits.  (Rounds 1-3 summed the BatchNorm / Gram statistics with
    f32 atomics: two eager runs differed by 3e-3 ... 2e-2 and this test could only bound the replay at 6e-2.  The statistics are now
    64-bit fixed point added with integer atomics - csrc/common.h, namespace fx - whose totals do not depend on arrival order.)"""
    from mhentropy_amd import harness
    torch.manual_seed(5)
    B, N = 256, 64
    model = harness.build_mhent(backbone="resnet50", tables=synth.mano_tables(0), compute_dtype=torch.bfloat16).cuda().train()
    xn, yn = synth.batch(9, B, image_size=256)
    x, y, z = torch.as_tensor(xn).cuda(), _t(yn), torch.as_tensor(synth.noise(9, N * B)).cuda()
    keys = ("th_norm", "bt_norm", "q_log_p_z_giv_y", "h_q_z_giv_i", "log_p")
    res = model.feat_extractor.res
    buffers = lambda: [b.clone() for b in (res.bn1.running_mean, res.bn1.running_var, res.layer4[2].bn3.running_mean, res.layer4[2].bn3.running_var)]
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    step = lambda: model.get_loss(x, y, mods=["uv"], N=N, noise=z)
    e1 = {k: v.clone() for k, v in step().items()}
    b1 = buffers()
    model.load_state_dict(state0)             # (the running statistics advance with every train-mode pass: same start for every run)
    e2 = {k: v.clone() for k, v in step().items()}
    b2 = buffers()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        og = step()
    model.load_state_dict(state0)
    g.replay()
    torch.cuda.synchronize()
    b3 = buffers()
    for k in keys:
        assert torch.isfinite(e1[k]).all(), k
        assert torch.equal(e2[k], e1[k]), (k, "second eager run", (e2[k].float() - e1[k].float()).abs().max().item())
        assert torch.equal(og[k], e1[k]), (k, "graph replay", (og[k].float() - e1[k].float()).abs().max().item())
    for u, v, w in zip(b1, b2, b3):
        assert torch.equal(u, v) and torch.equal(u, w), "BatchNorm running statistics"
    assert int(res.bn1.num_batches_tracked) == int(state0["feat_extractor.res.bn1.num_batches_tracked"]) + 1


@pytest.mark.parametrize("stats", ["stream", "gram"])
@pytest.mark.parametrize("training", [True, False])
def test_trunk_with_conv3_reevaluated_equals_the_stored_form(gpu_lib, training, stats):
    """MHE_FUSE_RECOMPUTE (layer1 / layer2 of ResNet-50 in bf16: conv3's raw output never written, csrc/conv_fuse.hip) against the same
    trunk writing and re-reading it.  stats = "stream": bn3's statistics from the statistics-only launch - same products in the same
    order, the pooled feature and the BatchNorm buffers agree to the grouping of the statistics' partial sums.  stats = "gram" (the default):
    statistics of the f32 products from the input's Gram matrix (csrc/conv_gram.hip) instead of the bf16-rounded outputs - bn3's batch
    statistics agree to ~1e-4; downstream a changed bf16 rounding is amplified by the later train-mode BatchNorms over this test's 128
    samples per channel, so the pooled feature is only held to the band by which bf16 storage itself moves it (measured 7e-2)."""
    from mhentropy_amd import resnet
    B, S = 8, 128
    sd = {k: torch.as_tensor(v) for k, v in synth.resnet_state(6, "resnet50").items()}
    x = torch.as_tensor(synth.batch(6, B, image_size=S)[0]).cuda()
    outs = []
    for re in (True, False):
        trunk = resnet.ResNetTrunk("resnet50", compute_dtype=torch.bfloat16)
        trunk.load_state_dict(sd)
        trunk = trunk.cuda().train(training)
        trunk.fuse_recompute, trunk.recompute_stats = re, stats
        f = trunk(x)
        outs.append((f.float().cpu(), trunk.layer1[0].bn3.running_var.cpu().clone(), trunk.layer1[0].bn3.running_mean.cpu().clone(),
                     int(trunk.layer1[2].bn3.num_batches_tracked)))
    (f1, rv1, rm1, n1), (f0, rv0, rm0, n0) = outs
    d = ((f1 - f0).abs().mean() / f0.abs().mean()).item()
    print(f"trunk feature, conv3 re-evaluated vs stored (training={training}, statistics={stats}): mean-rel {d:.2e}")
    # training: even another grouping of the statistics' partial sums (stats = "stream": otherwise identical arithmetic) can flip a bf16
    # rounding that the later BatchNorms over 128 samples amplify to ~5e-2 of the pooled feature (measured) - the tight check is on the
    # first bn3's buffers below; eval (running statistics): bit-level agreement
    assert d < (1e-6 if not training else 2e-1), d
    assert n1 == n0 == (1 if training else 0)
    tol = 1e-5 if stats == "stream" else 5e-4               # the first bn3 of the trunk: nothing upstream differs yet
    assert_close(rv1, rv0, tol, what="bn3 running_var")
    assert_close(rm1, rm0, tol, 1e-6, what="bn3 running_mean")


@pytest.mark.parametrize("training", [False, True])
def test_trunk_with_the_pool_inside_the_stem_equals_the_two_kernel_stem(gpu_lib, training):
    """MHE_STEM_POOL (csrc/stem_pool.hip; bf16, 256x256 images): layer1.0's conv1 / shortcut read the pooled RAW conv1 output with bn1 + ReLU
    on their operand load - the same values as stem -> bn1 -> relu -> maxpool (hand/network.py:54-61,110), so eval mode agrees to the last
    bit of the feature and train mode to the summation order of the statistics (the fused kernel sums conv1's outputs per strip, the
    two-kernel stem per tile: another f32 grouping, amplified like the recompute test's above)"""
    from mhentropy_amd import resnet
    sd = {k: torch.as_tensor(v) for k, v in synth.resnet_state(8, "resnet50").items()}
    sd["bn1.weight"] = sd["bn1.weight"].clone()
    sd["bn1.weight"][::4] *= -1.0                                   # negative BatchNorm scales: the window minimum is the one that matters
    x = torch.as_tensor(synth.batch(8, 4, image_size=256)[0]).cuda()
    outs = []
    for fused in (True, False):
        trunk = resnet.ResNetTrunk("resnet50", compute_dtype=torch.bfloat16)
        trunk.load_state_dict(sd)
        trunk = trunk.cuda().train(training)
        trunk.stem_pool_fused = fused
        outs.append((trunk(x).float().cpu(), trunk.bn1.running_var.cpu().clone(), trunk.layer1[0].bn1.running_mean.cpu().clone()))
    (f1, rv1, rm1), (f0, rv0, rm0) = outs
    d = ((f1 - f0).abs().mean() / f0.abs().mean()).item()
    print(f"trunk feature, pool inside the stem vs two kernels (training={training}): mean-rel {d:.2e}")
    assert d < (2e-1 if training else 1e-6), d
    assert_close(rv1, rv0, 1e-5, what="bn1 running_var")
    assert_close(rm1, rm0, 1e-4, 1e-6, what="layer1.0.bn1 running_mean (first consumer of the pooled output)")


@pytest.mark.parametrize("arch,dt,S", [("resnet50", torch.bfloat16, 128), ("resnet50", torch.float32, 96), ("resnet18", torch.bfloat16, 96)])
@pytest.mark.parametrize("training", [True, False])
def test_trunk_with_the_last_tail_inside_the_average_pool_equals_the_two_launches(gpu_lib, arch, dt, S, training):
    """MHE_FUSE_POOL (csrc/conv.hip, bn_act_avgpool_kernel): the last block's relu(bn(y) + identity) evaluated inside the global average pool
    (torchvision ResNet.forward: layer4 -> avgpool; hand/network.py:54-61,110) - rounded to the storage type before it is summed, summed in
    the pool kernel's order: the encoder feature equals the two launches' BIT FOR BIT, in both modes and both storage types"""
    from mhentropy_amd import resnet
    sd = {k: torch.as_tensor(v) for k, v in synth.resnet_state(12, arch).items()}
    x = torch.as_tensor(synth.batch(12, 6, image_size=S)[0]).cuda()
    outs = []
    for fused in (True, False):
        trunk = resnet.ResNetTrunk(arch, compute_dtype=dt)
        trunk.load_state_dict(sd)
        trunk = trunk.cuda().train(training)
        trunk.fuse_pool = fused
        outs.append(trunk(x))
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
