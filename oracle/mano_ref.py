"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the MANO forward the reference runs
(/root/reference/hand/manopth/manolayer.py:110-274 through the wrapper
/root/reference/hand/ManoLayer.py:45-60) for the configuration MHEnt builds:
use_pca=True, ncomps=45, flat_hand_mean=False, center_idx=9, side='right',
root_rot_mode='axisang', no translation (reference hand/CrossModalHand.py:72-74,
hand/ManoLayer.py:19-21).  Tables are a dict keyed by the reference's buffer
names (`th_shapedirs`, `th_posedirs`, ...).  Pinned by tests/golden/mano_*.npz.
"""
import torch

PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)
TIP_VERTS_RIGHT = (745, 317, 444, 556, 673)          # manolayer.py:251
JOINT_REORDER = (0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20)  # :260
FREIHAND2RHD = (0, 4, 3, 2, 1, 8, 7, 6, 5, 12, 11, 10, 9, 16, 15, 14, 13, 20, 19, 18, 17)  # utils.py:15
WRAPPER_TIP_VERTS = {4: 744, 8: 320, 12: 443, 16: 555, 20: 672}      # ManoLayer.py:112-118
WRAPPER_JOINT_MAP = {0: 0, 1: 5, 2: 6, 3: 7, 4: 9, 5: 10, 6: 11, 7: 17, 8: 18, 9: 19,
                     10: 13, 11: 14, 12: 15, 13: 1, 14: 2, 15: 3}   # ManoLayer.py:121-126


def tables_from_numpy(t, dtype=torch.float32):
    """numpy tables (mhentropy_amd.synth.mano_tables) -> reference buffer names
    (manolayer.py:69-101)."""
    f = lambda a: torch.as_tensor(a).to(dtype)
    return {
        "th_betas": f(t["betas"])[None],
        "th_shapedirs": f(t["shapedirs"]),
        "th_posedirs": f(t["posedirs"]),
        "th_v_template": f(t["v_template"])[None],
        "th_J_regressor": f(t["J_regressor"]),
        "th_weights": f(t["weights"]),
        "th_faces": torch.as_tensor(t["faces"]).long(),
        "th_hands_mean": f(t["hands_mean"])[None],
        "th_comps": f(t["hands_components"]),
        "th_selected_comps": f(t["hands_components"][:45]),
    }


def rodrigues(aa):
    """axis-angle (N,3) -> rotation (N,3,3) through a unit quaternion.
    reference rodrigues_layer.py:43-54 (batch_rodrigues) and :15-40 (quat2mat);
    note the +1e-8 added to every component INSIDE the norm (:45)."""
    angle = torch.norm(aa + 1e-8, p=2, dim=1, keepdim=True)
    axis = aa / angle
    half = angle * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * axis], 1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    r = torch.stack([
        w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], 1)
    return r.view(-1, 3, 3)


def mano_forward(tb, theta, beta):
    """theta (R,48) = [root axis-angle(3), PCA coefficients(45)], beta (R,10)
    -> verts (R,778,3) mm, joints (R,21,3) mm, both centred on joint 9.
    Steps follow manolayer.py: PCA :131-143, rotations :144-149 (tensutils.py:6-12),
    shape blend + joint regression :181-184, pose blend :187-188, kinematic chain
    :193-229, skinning :231-246, tips/reorder :250-260, centring + mm :262-273."""
    R = theta.shape[0]
    full_pose = torch.cat([theta[:, :3], tb["th_hands_mean"] + theta[:, 3:48].mm(tb["th_selected_comps"])], 1)
    rots = rodrigues(full_pose.reshape(-1, 3)).view(R, 16, 3, 3)
    eye = torch.eye(3, dtype=theta.dtype)
    pose_map = (rots[:, 1:] - eye).reshape(R, 135)

    v_shaped = torch.matmul(tb["th_shapedirs"], beta.t()).permute(2, 0, 1) + tb["th_v_template"]
    th_j = torch.matmul(tb["th_J_regressor"], v_shaped)                       # (R,16,3)
    v_posed = v_shaped + torch.matmul(tb["th_posedirs"], pose_map.t()).permute(2, 0, 1)

    def rigid(rot, trans):       # (R,3,3),(R,3) -> (R,4,4)   tensutils.py:15-22
        top = torch.cat([rot, trans.unsqueeze(2)], 2)
        bot = theta.new_tensor([0.0, 0.0, 0.0, 1.0]).view(1, 1, 4).repeat(R, 1, 1)
        return torch.cat([top, bot], 1)

    G = [None] * 16
    G[0] = rigid(rots[:, 0], th_j[:, 0])
    for j in range(1, 16):
        p = PARENTS[j]
        G[j] = torch.matmul(G[p], rigid(rots[:, j], th_j[:, j] - th_j[:, p]))
    G = torch.stack(G, 1)                                                    # (R,16,4,4)

    j_h = torch.cat([th_j, th_j.new_zeros(R, 16, 1)], 2)
    corr = torch.matmul(G, j_h.unsqueeze(3))                                 # (R,16,4,1)
    G_rest = G - torch.cat([corr.new_zeros(R, 16, 4, 3), corr], 3)           # manolayer.py:231-234
    T = torch.matmul(G_rest.permute(0, 2, 3, 1), tb["th_weights"].t())       # (R,4,4,778)
    rest_h = torch.cat([v_posed.transpose(2, 1), theta.new_ones(R, 1, v_posed.shape[1])], 1)
    verts = (T * rest_h.unsqueeze(1)).sum(2).transpose(2, 1)[:, :, :3]

    jtr = G[:, :, :3, 3]
    jtr = torch.cat([jtr, verts[:, list(TIP_VERTS_RIGHT)]], 1)[:, list(JOINT_REORDER)]
    center = jtr[:, 9].unsqueeze(1)
    return (verts - center) * 1000, (jtr - center) * 1000


def wrapper_forward(tb, theta, beta, skeidx="RHD"):
    """reference hand/ManoLayer.py:45-60: returns mesh, mano_joints (reordered to the
    RHD skeleton) and `joints` re-regressed from the mesh (:141-148, :108-139)."""
    verts, mano_joints = mano_forward(tb, theta, beta)
    jr = tb["th_J_regressor"].t()
    reg = torch.stack([verts[:, :, c].matmul(jr) for c in range(3)], 2)      # (R,16,3)
    kp = [None] * 21
    for src, dst in WRAPPER_JOINT_MAP.items():
        kp[dst] = reg[:, src]
    for dst, vid in WRAPPER_TIP_VERTS.items():
        kp[dst] = verts[:, vid]
    joints = torch.stack(kp, 1)
    if skeidx == "RHD":
        joints = joints[:, list(FREIHAND2RHD)]
        mano_joints = mano_joints[:, list(FREIHAND2RHD)]
    return {"beta": beta, "theta": theta, "mesh": verts, "joints": joints, "mano_joints": mano_joints}


def normalize_pose3d(pose3d, root_idx=12, norm_idx=11):
    """reference hand/utils.py:46-66 with return_st=True."""
    root = pose3d[:, root_idx].clone().view(-1, 1, 3)
    rel = pose3d - root
    bone = torch.sqrt(torch.sum(rel[:, norm_idx] ** 2, -1)).view(-1, 1, 1)
    return rel / bone, root, bone[:, 0, 0]


def orth_proj(joint, scale, trans, image_size=256, inv_norm=True):
    """reference hand/ManoLayer.py:150-165."""
    out = scale[:, None, :] * joint[:, :, :2] + trans[:, None, :]
    if inv_norm:
        out = (out + 1.0) / 2.0 * image_size
    return out
