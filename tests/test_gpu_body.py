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


@pytest.mark.parametrize("R", [1, 5, 19])
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
