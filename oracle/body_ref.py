"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of linear-blend skinning for a body model of ARBITRARY size (J joints, V vertices, nb shape and 9(J-1)
pose-blend coefficients, any kinematic tree), taking rotation matrices - what a ProHMR-style 6D pose head produces
(reference README.md:26-42; rot6d at hand/manopth/rot6d.py:4-24).  The arithmetic is the one the reference's vendored
manopth runs for the hand (hand/manopth/manolayer.py:181-246: shape blend, joint regression, pose-corrective blend,
kinematic chain, rest-pose removal, weighted transforms), only the sizes differ.

PINNED at the MANO sizes (tests/test_oracle_golden.py: with MANO tables and the hand's rotations this function reproduces
oracle/mano_ref.mano_forward, itself pinned by reference fixtures); the SMPL tables themselves (6,890 x 24) are out of tree
(licence), so at body size the inputs are synthetic and PARITY with ProHMR's SMPL layer is UNPINNED."""
import torch


def lbs(tb, rotmats, betas):
    """tb: dict v_template (V,3), shapedirs (V,3,nb), posedirs (V,3,9(J-1)), J_regressor (J,V), weights (V,J), parents (J,);
    rotmats (R,J,3,3), betas (R,nb) -> verts (R,V,3), posed joints (R,J,3)"""
    R, J = rotmats.shape[:2]
    parents = [int(p) for p in tb["parents"]]
    eye = torch.eye(3, dtype=rotmats.dtype)
    pose_map = (rotmats[:, 1:] - eye).reshape(R, 9 * (J - 1))
    v_shaped = torch.matmul(tb["shapedirs"], betas.t()).permute(2, 0, 1) + tb["v_template"]          # manolayer.py:181-183
    j_rest = torch.matmul(tb["J_regressor"], v_shaped)                                               # :184
    v_posed = v_shaped + torch.matmul(tb["posedirs"], pose_map.t()).permute(2, 0, 1)                 # :187-188

    def rigid(rot, trans):
        top = torch.cat([rot, trans.unsqueeze(2)], 2)
        bot = rotmats.new_tensor([0.0, 0.0, 0.0, 1.0]).view(1, 1, 4).repeat(R, 1, 1)
        return torch.cat([top, bot], 1)

    G = [None] * J
    G[0] = rigid(rotmats[:, 0], j_rest[:, 0])
    for j in range(1, J):                                                                            # :193-229
        G[j] = torch.matmul(G[parents[j]], rigid(rotmats[:, j], j_rest[:, j] - j_rest[:, parents[j]]))
    G = torch.stack(G, 1)
    j_h = torch.cat([j_rest, j_rest.new_zeros(R, J, 1)], 2)
    corr = torch.matmul(G, j_h.unsqueeze(3))
    G_rest = G - torch.cat([corr.new_zeros(R, J, 4, 3), corr], 3)                                    # :231-234
    T = torch.matmul(G_rest.permute(0, 2, 3, 1), tb["weights"].t())                                  # (R,4,4,V)  :236
    rest_h = torch.cat([v_posed.transpose(2, 1), rotmats.new_ones(R, 1, v_posed.shape[1])], 1)
    verts = (T * rest_h.unsqueeze(1)).sum(2).transpose(2, 1)[:, :, :3]                               # :245-246
    return verts, G[:, :, :3, 3]
