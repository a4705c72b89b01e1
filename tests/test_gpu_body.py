"""GPU parity of the body-model path: 6D rotation -> R against vectors generated from the reference's
own hand/manopth/rot6d.py; size-generic linear-blend skinning against the reference-pinned hand mesh at the MANO sizes and
against the size-generic oracle at SMPL's sizes (synthetic tables: the SMPL model and ProHMR's classes are out of tree, so parity
with ProHMR itself is unpinned); the 144-D flow head's call surface and its hypothesis-sliced decode."""
import numpy as np
import pytest
import torch

from conftest import load_golden, assert_close
from mhentropy_amd import synth

pytestmark = pytest.mark.gpu


def test_rot6d_matches_reference_vectors(gpu_lib):
    from mhentropy_amd import body
    g = load_golden("rot6d")
    p = torch.as_tensor(g["poses"]).cuda()
    assert_close(body.rot6d_to_rotmat(p).cpu(), g["R"], 1e-6, what="R (rot6d.py:4-24)")
    assert_close(body.rot6d_to_rotmat(p, robust=True).cpu(), g["R_robust"], 1e-6, what="R (rot6d.py:26-51)")
    gp = body.rot6d_to_rotmat_bwd(p, torch.as_tensor(g["w"]).cuda())
    assert_close(gp.cpu(), g["grad_poses"], 1e-5, what="reverse: d sum(w R) / d poses (reference autograd)")
    # batched shapes as the head uses them: (B, K, 24, 6)
    q = torch.as_tensor(g["poses"][:48]).reshape(2, 1, 24, 6).cuda()
    assert body.rot6d_to_rotmat(q).shape == (2, 1, 24, 3, 3)


def test_generic_lbs_at_mano_size_matches_reference_mesh(gpu_lib):
    """BodyLayer on MANO's tables / tree / rotations == the mesh the reference's manopth returned (tests/golden/mano.npz)"""
    from mhentropy_amd import body
    from oracle import mano_ref
    g = load_golden("mano")
    t = synth.mano_tables(int(g["table_seed"]))
    tb = mano_ref.tables_from_numpy(t)
    theta, beta = torch.as_tensor(g["theta"]), torch.as_tensor(g["beta"])
    full_pose = torch.cat([theta[:, :3], tb["th_hands_mean"] + theta[:, 3:48].mm(tb["th_selected_comps"])], 1)
    rots = mano_ref.rodrigues(full_pose.reshape(-1, 3)).view(-1, 16, 3, 3)
    layer = body.BodyLayer({"v_template": t["v_template"], "shapedirs": t["shapedirs"], "posedirs": t["posedirs"],
                            "J_regressor": t["J_regressor"], "weights": t["weights"], "parents": np.asarray(mano_ref.PARENTS)}).cuda()
    out = layer(beta.cuda(), rotmats=rots.cuda())
    centre = out["joints"][:, mano_ref.JOINT_REORDER[9]].unsqueeze(1)
    assert_close(((out["vertices"] - centre) * 1000).cpu(), g["mesh"], 1e-4, what="mesh (mm, centred on joint 9)")


@pytest.mark.parametrize("R", [1, 5, 19, 70])
def test_generic_lbs_at_smpl_size_matches_oracle(gpu_lib, R):
    from mhentropy_amd import body
    from oracle import body_ref, rot6d_ref
    t = body.synthetic_body_tables(1)
    layer = body.BodyLayer(t).cuda()
    rng = np.random.default_rng(R)
    p6 = torch.as_tensor(rng.normal(0, 1, (R, 144)).astype(np.float32))
    betas = torch.as_tensor(rng.normal(0, 1, (R, 10)).astype(np.float32))
    out = layer(betas.cuda(), pose6d=p6.cuda())
    tb = {k: torch.as_tensor(v) for k, v in t.items()}
    rm = rot6d_ref.rotation_from_ortho6d(p6.view(R, 24, 6))
    verts, joints = body_ref.lbs(tb, rm, betas)
    assert out["vertices"].shape == (R, 6890, 3) and out["joints"].shape == (R, 24, 3)
    assert_close(out["rotmats"].cpu(), rm, 1e-6, what="rotation matrices")
    assert_close(out["joints"].cpu(), joints, 1e-4, what="posed joints")
    assert_close(out["vertices"].cpu(), verts, 1e-4, what="vertices")
    # round 5: the skinning runs on the matrix cores from bf16 pieces of the f32 operands (csrc/lbs_skin.hip) - against the f64 oracle it stays
    # within three times the f32 oracle's own distance (floor 2e-6 of the mesh's extent)
    tb64 = {k: (torch.as_tensor(v).double() if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v)) for k, v in t.items()}
    v64, _ = body_ref.lbs(tb64, rm.double(), betas.double())
    ext = float(v64.abs().max())
    e_gpu, e_f32 = float((out["vertices"].cpu().double() - v64).abs().max()) / ext, float((verts.double() - v64).abs().max()) / ext
    print(f"SMPL-size mesh R={R}: max error / extent  HIP {e_gpu:.2e}   f32 oracle {e_f32:.2e}")
    assert e_gpu <= max(3 * e_f32, 2e-6), (e_gpu, e_f32)
    big = torch.full((R + 40, 6890, 3), -7.0, device="cuda")          # rows past R are not stored
    from mhentropy_amd import _lib, ops
    L = _lib.lib()
    if L.mhe_lbs_skin_mfma_supported(R, 24, 10, 6890, layer.VP):
        ws = torch.empty(L.mhe_lbs_workspace_floats(R, 24, 10), device="cuda")
        jt = torch.empty(R, 24, 3, device="cuda")
        rmd, bd = out["rotmats"].contiguous(), betas.cuda().contiguous()
        ops.check(L.mhe_lbs_pose_f32(ops._ptr(rmd), ops._ptr(bd), ops._ptr(layer._jt), ops._ptr(layer._jsd), ops._ptr(layer.parents), ops._ptr(ws), ops._ptr(jt),
                                     R, 24, 10, ops._stream()), "pose")
        ops.check(L.mhe_lbs_skin_mfma_f32(ops._ptr(ws), ops._ptr(layer._split_tables(torch.device("cuda", 0))), ops._ptr(big), R, 24, 10, 6890, layer.VP, 1.0,
                                          ops._stream()), "skin")
        assert_close(big[:R].cpu(), out["vertices"].cpu(), 1e-5, what="direct call")          # (bit-equal unless MHE_LBS_MFMA=0 chose the other kernel above)
        assert bool((big[R:] == -7.0).all())


def test_body_flow_head_and_hypothesis_slices(gpu_lib):
    """`flow(feats, K)` -> K poses with log-probabilities, decoded to meshes; decoding hypothesis slices separately (what a
    hypothesis-sharded rank does) gives exactly the rows of the full decode"""
    from mhentropy_amd import body
    B, K = 2, 6
    head = body.BodyFlowHead(body.synthetic_body_tables(2), context_features=256, hidden=128, num_layers=2, num_blocks=1)
    sd = {k: torch.as_tensor(v) for k, v in synth.glow_state(5, 144, 128, 2, 1, 256).items()}
    head.flow.load_state_dict(sd, strict=False)
    head = head.cuda().eval()
    rng = np.random.default_rng(3)
    feats = torch.as_tensor(rng.normal(0, 0.5, (B, 256)).astype(np.float32)).cuda()
    noise = torch.as_tensor(rng.normal(0, 1, (B, K, 144)).astype(np.float32)).cuda()
    betas = torch.as_tensor(rng.normal(0, 1, (B, 10)).astype(np.float32)).cuda()
    full = head(feats, K, betas=betas, noise=noise)
    assert full["pose6d"].shape == (B, K, 144) and full["log_prob"].shape == (B, K) and full["vertices"].shape == (B, K, 6890, 3)
    assert torch.isfinite(full["vertices"]).all() and torch.isfinite(full["log_prob"]).all()
    for lo, hi in ((0, 3), (3, 6)):
        part = head(feats, K, betas=betas, noise=noise, hyp_slice=(lo, hi))
        assert torch.equal(part["vertices"], full["vertices"][:, lo:hi]) and torch.equal(part["joints"], full["joints"][:, lo:hi])


def test_c4_per_gpu_size_properties(gpu_lib):
    """config C4 (BASELINE.json configs[3]: ProHMR body flow, global batch 1,024 x K = 128 on 8 GPUs -> B = 128 images per GPU,
    R = 16,384 hypotheses; Glow features 144 / hidden 1,024 / 4 layers x 2 blocks / context 2,048; SMPL-sized body: 24 joints,
    6,890 vertices).  No oracle at this size (the flow is parity-unpinned anyway): properties -
    (i) density consistency: log_prob(sample) evaluated by the inverse pass equals the log-probability returned with the sample,
        and the inverse pass recovers the noise;
    (ii) a hypothesis whose 6D pose is the identity decodes to the shaped template (skinning with identity transforms);
    (iii) hypothesis slices decode to exactly the rows of the full decode; everything finite."""
    from mhentropy_amd import body
    B, K = 128, 128
    tables = body.synthetic_body_tables(4)
    head = body.BodyFlowHead(tables)                                   # ProHMR sizes are the defaults
    torch.manual_seed(7)
    for p in head.flow.parameters():                                   # small random weights (the class initialises like nflows)
        if p.dim() > 1:
            torch.nn.init.normal_(p, 0, 0.3 / p.shape[1] ** 0.5)
    head = head.cuda().eval()
    gen = torch.Generator(device="cuda").manual_seed(1)
    feats = torch.randn(B, 2048, device="cuda", generator=gen) * 0.5
    noise = torch.randn(B, K, 144, device="cuda", generator=gen)
    betas = torch.randn(B, 10, device="cuda", generator=gen)
    full = head(feats, K, betas=betas, noise=noise)
    assert full["vertices"].shape == (B, K, 6890, 3) and full["joints"].shape == (B, K, 24, 3)
    for k in ("pose6d", "log_prob", "vertices", "joints"):
        assert torch.isfinite(full[k]).all(), k
    # (i) rows of log_prob are sample-major with B context rows: row r = n*B + b
    x_sm = full["pose6d"].permute(1, 0, 2).reshape(K * B, 144).contiguous()
    lp, z = head.flow.log_prob(x_sm, context=feats)
    assert_close(lp.view(K, B).t().cpu(), full["log_prob"].cpu(), 2e-3, what="log_prob(sample) == log-prob of the sampling pass")
    assert_close(z.view(K, B, 144).permute(1, 0, 2).cpu(), noise.cpu(), 2e-2, what="inverse pass recovers the noise (bf16 hidden products)")
    # (ii) identity pose -> shaped template
    eye6 = torch.tensor([1., 0, 0, 0, 1, 0], device="cuda").repeat(24)
    rest = head.body(betas[:4].contiguous(), pose6d=eye6.repeat(4, 1).contiguous())
    tpl = torch.as_tensor(tables["v_template"]).cuda() + torch.einsum("vck,bk->bvc", torch.as_tensor(tables["shapedirs"]).cuda().float(), betas[:4])
    assert_close(rest["vertices"].cpu(), tpl.cpu(), 1e-5, what="identity pose = shaped template")
    # (iii) a 1/8 hypothesis slice (what one of 8 hypothesis-sharded ranks decodes)
    part = head(feats, K, betas=betas, noise=noise, hyp_slice=(32, 48))
    assert torch.equal(part["vertices"], full["vertices"][:, 32:48]) and torch.equal(part["joints"], full["joints"][:, 32:48])
