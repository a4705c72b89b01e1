"""CPU: the oracle restatement against the golden vectors captured from the
reference itself (oracle/gen_golden.py).  These pin the oracle; the GPU tests
then compare the HIP path with the oracle / the same vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden, assert_close
from mhentropy_amd import synth
from oracle import flows_ref, mano_ref, network_ref, criteria_ref


def _t(d):
    return {k: torch.as_tensor(v) for k, v in d.items()}


@pytest.mark.parametrize("tag", ["small", "shipped"])
def test_flow_oracle_matches_reference_vectors(tag):
    g = load_golden(f"flow_{tag}")
    sd = _t(synth.flow_state(int(g["seed"]), 45, int(g["cond_dim"]), (int(g["h"]),) * 2, int(g["steps"])))
    z0, feat = torch.as_tensor(g["z0"]), torch.as_tensor(g["feat"])
    with torch.no_grad():
        x, tot = flows_ref.forward_p_logdet(sd, z0, feat)
        zb, ld = flows_ref.backward_p(sd, torch.as_tensor(g["x"]), feat)
        lp = flows_ref.log_prob(sd, torch.as_tensor(g["x"]), feat)
    assert_close(x, g["x"], 1e-6, what="forward_p")
    assert_close(zb, g["z_back"], 1e-6, what="backward_p z")
    assert_close(ld, g["log_det"], 1e-6, what="log_det")
    assert_close(lp, g["log_prob"], 1e-6, what="log_prob")
    # invariants the reference itself satisfies
    assert_close(zb, g["z0"], 1e-5, what="backward(forward(z)) == z")
    assert_close(flows_ref.std_normal_logprob(z0) - tot, g["log_prob"], 1e-5, what="fused log q")


def test_mano_oracle_matches_reference_vectors():
    g = load_golden("mano")
    tb = mano_ref.tables_from_numpy(synth.mano_tables(int(g["table_seed"])))
    out = mano_ref.wrapper_forward(tb, torch.as_tensor(g["theta"]), torch.as_tensor(g["beta"]))
    for k in ("mesh", "mano_joints", "joints"):
        assert_close(out[k], g[k], 1e-6, what=k)


def test_mano_invariants():
    tb = mano_ref.tables_from_numpy(synth.mano_tables(0))
    R = mano_ref.rodrigues(torch.randn(64, 3, dtype=torch.float64))
    eye = torch.eye(3, dtype=torch.float64).expand(64, 3, 3)
    assert_close(R @ R.transpose(1, 2), eye, 1e-12, 1e-12, "orthonormal")
    assert_close(torch.linalg.det(R), np.ones(64), 1e-12, 1e-12, "det +1")
    # zero pose (flat mean) => LBS leaves v_shaped untouched up to the centring
    tb0 = dict(tb)
    tb0["th_hands_mean"] = torch.zeros_like(tb["th_hands_mean"])
    beta = torch.randn(3, 10) * 0.02
    verts, jtr = mano_ref.mano_forward(tb0, torch.zeros(3, 48), beta)
    v_shaped = torch.matmul(tb["th_shapedirs"], beta.t()).permute(2, 0, 1) + tb["th_v_template"]
    assert_close(verts - verts[:, :1], (v_shaped - v_shaped[:, :1]) * 1000, 1e-4, what="zero pose")


@pytest.mark.parametrize("tag", ["small", "shipped"])
def test_mhent_oracle_matches_reference_vectors(tag):
    g = load_golden(f"mhent_{tag}")
    seed, h, steps, B = int(g["seed"]), int(g["h"]), int(g["steps"]), int(g["B"])
    sdn = {"q_z_giv_i." + k: v for k, v in synth.flow_state(seed, 45, 512, (h, h), steps).items()}
    sdn.update(synth.head_state(seed, 2048, 512, 16))
    sd = _t(sdn)
    tb = mano_ref.tables_from_numpy(synth.mano_tables(0))
    y = _t({k[2:]: v for k, v in g.items() if k.startswith("y_")})
    feat = torch.nn.functional.linear(torch.as_tensor(g["trunk"]), sd["feat_extractor.l1.0.weight"],
                                      sd["feat_extractor.l1.0.bias"])
    assert_close(feat, g["feat"], 1e-6, what="feat")
    N = int(g["N_loss"])
    with torch.no_grad():
        out = network_ref.reverse_kld(sd, tb, feat, y, torch.as_tensor(g["z0_loss"]), N)
        terms = network_ref.forward_log_p(tb, torch.as_tensor(g["z_loss"]), y, N)
    for k in ("th_norm", "bt_norm", "q_log_p_z_giv_y", "h_q_z_giv_i", "log_p"):
        assert_close(out[k], g["loss_" + k], 1e-6, what=k)
    for k in ("log_p_uv_giv_z", "log_p_th3", "log_p_th45", "log_p_bt"):
        assert_close(terms[k], g["terms_" + k], 1e-6, 1e-12, what=k)
    with torch.no_grad():
        s = network_ref.sample(sd, tb, feat, torch.as_tensor(g["z0_sample"]), 4)
    for k in ("th_bt", "logs_t", "verts", "xyz", "uv"):
        assert_close(s[k], g["sample_" + k], 1e-6, what="sample." + k)
    with torch.no_grad():
        tk = network_ref.sample(sd, tb, feat, torch.as_tensor(g["z0_topk"]), 6, N_quant=3)
    for k in ("th_bt", "logs_t", "verts", "xyz", "uv"):
        assert_close(tk[k], g["topk_" + k], 1e-6, what="topk." + k)
    o = {"log_p": torch.as_tensor(g["loss_log_p"]), "xyz": torch.as_tensor(g["sample_xyz"]),
         "uv": torch.as_tensor(g["sample_uv"])}
    tot, _, met = criteria_ref.mhent_loss(o, y)
    assert_close(tot, g["criterion_total"], 1e-6, what="criterion")
    for k, v in met.items():
        assert_close(v, g["metric_" + k], 1e-6, what=k)


def test_priors_oracle_outside_their_supports_matches_reference_vectors():
    """soft-support priors far outside their supports (reference hand/network.py:155-165,429-435), pinned by values the
    reference's own _forward_log_p returned (oracle/gen_golden.py:gen_priors)"""
    g = load_golden("priors_outside")
    tb = mano_ref.tables_from_numpy(synth.mano_tables(0))
    y = _t({k[2:]: v for k, v in g.items() if k.startswith("y_")})
    with torch.no_grad():
        out = network_ref.forward_log_p(tb, torch.as_tensor(g["z"]), y, int(g["N"]))
    for k in ("log_p_uv_giv_z", "log_p_th3", "log_p_th45", "log_p_bt", "log_p"):
        assert_close(out[k], g["terms_" + k], 1e-6, what=k)
    assert g["terms_log_p_th3"].min() < -10 and g["terms_log_p_bt"].min() < -100 and g["terms_log_p_th45"].min() < -1000


def test_rot6d_oracle_matches_reference_vectors():
    from oracle import rot6d_ref
    g = load_golden("rot6d")
    p = torch.as_tensor(g["poses"])
    assert_close(rot6d_ref.rotation_from_ortho6d(p), g["R"], 1e-6, what="R")
    assert_close(rot6d_ref.rotation_from_ortho6d_robust(p), g["R_robust"], 1e-6, what="R (robust form)")
    R = rot6d_ref.rotation_from_ortho6d(p.double())
    assert_close(R.transpose(1, 2) @ R, torch.eye(3).expand(len(p), 3, 3), 1e-12, 1e-12, what="orthonormal")
    assert_close(torch.linalg.det(R), np.ones(len(p)), 1e-12, 1e-12, what="det +1")
    pr = p.clone().requires_grad_(True)
    (rot6d_ref.rotation_from_ortho6d(pr) * torch.as_tensor(g["w"])).sum().backward()
    assert_close(pr.grad, g["grad_poses"], 1e-5, what="d/d poses")


def test_body_lbs_oracle_at_mano_size_is_the_pinned_hand_forward():
    """the size-generic skinning restatement (oracle/body_ref.py, row f1) reproduces the reference-pinned hand forward when fed the
    MANO tables, the hand's kinematic tree and the hand's rotations (reference hand/manopth/manolayer.py:181-246)"""
    from oracle import body_ref
    g = load_golden("mano")
    tb = mano_ref.tables_from_numpy(synth.mano_tables(int(g["table_seed"])))
    theta, beta = torch.as_tensor(g["theta"]), torch.as_tensor(g["beta"])
    full_pose = torch.cat([theta[:, :3], tb["th_hands_mean"] + theta[:, 3:48].mm(tb["th_selected_comps"])], 1)
    rots = mano_ref.rodrigues(full_pose.reshape(-1, 3)).view(-1, 16, 3, 3)
    bt = {"v_template": tb["th_v_template"][0], "shapedirs": tb["th_shapedirs"], "posedirs": tb["th_posedirs"],
          "J_regressor": tb["th_J_regressor"], "weights": tb["th_weights"], "parents": mano_ref.PARENTS}
    verts, joints = body_ref.lbs(bt, rots, beta)
    centre = joints[:, mano_ref.JOINT_REORDER[9]].unsqueeze(1)             # manolayer.py:262-266
    assert_close((verts - centre) * 1000, g["mesh"], 1e-6, what="mesh")


@pytest.mark.parametrize("case", range(6))
def test_ho3d_pipeline_oracle_matches_reference_fixtures(case):
    """row f4: oracle/ho3d_ref.getitem against what the reference's own Generate_ho3d_uv.__getitem__ returned on the same synthetic
    decoded sample and the same random draws (fixtures by oracle/gen_golden.py:gen_ho3d; crops inside and across the image
    border, with and without augmentation)"""
    from conftest import check_ho3d_against_fixture
    from oracle import ho3d_ref
    g = load_golden(f"ho3d_{case}")
    smp = synth.ho3d_sample(int(g["seed"]), tuple(float(v) for v in g["offset"]))
    prm = synth.ho3d_aug_params(int(g["aug_seed"])) if int(g["aug"]) else None
    img, tgt = ho3d_ref.getitem(smp, prm)
    check_ho3d_against_fixture(img, tgt, g, f"ho3d case {case}")
