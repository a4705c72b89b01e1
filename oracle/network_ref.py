"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the MHEnt loss and sampling path
(/root/reference/hand/network.py) with the hypothesis count N and the base noise
as explicit arguments (the reference hard-codes N=10, network.py:780, and draws
noise from the global generator, flows.py:339).  Works on one flat state_dict
with the reference's key names (`feat_extractor.*`, `q_z_giv_i.*`, `det_head.*`).
Pinned by tests/golden/mhent_*.npz.
"""
import math
import torch
import torch.nn.functional as F

from . import flows_ref, mano_ref, resnet_ref

LAPLACE_B = 0.03          # network.py:392 (b_init = data_prior_cfg['b_2d']), ho3d.yaml:44
TH45_ALPHA = 50.0         # network.py:427-429, ho3d.yaml:41
ROOT_IDX, NORM_IDX = 12, 11   # network.py:477-478 ('ho3d')


def sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def encoder(sd, x, arch="resnet50", training=True, bf16_storage=False):
    """BasicEnc.forward, network.py:96-140: trunk, then mn = l1(feat).  The caller keeps
    only mn (network.py:779); l2/exp/epsilon are dead for MHEnt and not restated.
    bf16_storage=True: the trunk with the bf16 performance mode's rounding points (resnet_ref.forward_bf16_storage: every stored
    tensor rounded to bf16 - and, under autograd, every gradient passing a storage point -, all arithmetic f32): the yardstick for what
    bf16 STORAGE alone does to a result, with exact arithmetic everywhere else."""
    fwd = resnet_ref.forward_bf16_storage if bf16_storage else resnet_ref.forward
    trunk = fwd(sub(sd, "feat_extractor.res."), x, arch=arch, training=training)
    return F.linear(trunk, sd["feat_extractor.l1.0.weight"], sd["feat_extractor.l1.0.bias"])


def det_head(sd, feat):
    """network.py:380-383: Linear(512,512) -> ReLU -> Linear(512,16)."""
    h = F.relu(F.linear(feat, sd["det_head.0.weight"], sd["det_head.0.bias"]))
    return F.linear(h, sd["det_head.2.weight"], sd["det_head.2.bias"])


def combine_z(z_det, th45):
    """network.py:703-717 with zdims order th3,th45,bt,logs,t and det order th3,bt,logs,t."""
    return torch.cat([z_det[:, 0:3], th45, z_det[:, 3:13], z_det[:, 13:14], z_det[:, 14:16]], 1)


def sample_q(sd, feat, z0, N):
    """_sample_q_z_giv_i, network.py:719-758 (RealNVP branch).  z0 is (N*B,45), already
    scaled by temp; rows are sample-major (feat.repeat(N,1), network.py:734)."""
    th45 = flows_ref.sample(sub(sd, "q_z_giv_i."), z0, feat.repeat(N, 1))
    return combine_z(det_head(sd, feat).repeat(N, 1), th45)


def approx_uniform_rec(x, a, b, alpha):
    """network.py:155-158."""
    return -(alpha * F.relu((x - (a + b) / 2.0).abs() / ((b - a) / 2.0) - 1.0) ** 2).sum(1)


def approx_uniform_ball(x, radius, alpha):
    """network.py:159-163 with centre 0."""
    r = x.norm(p=2, dim=-1)
    return -alpha * F.relu(r / radius - 1.0) ** 2


def laplace_log_prob(x, mu, weights, b=LAPLACE_B):
    """_Laplace.log_prob, network.py:233-258, const b."""
    b = torch.tensor([b], dtype=x.dtype)
    return ((weights == 1.0) * (-(F.relu((x - mu).abs() - 1e-4) + 1e-4) / b - torch.log(2 * b))).flatten(1).sum(1)


def decode(tb, z, image_size=256, inv_norm=False):
    """_th_bt_product, network.py:541-558 (mods=['uv'], render=[]): MANO decode,
    normalise on root 12 / bone 11 (network.py:466-483), orthographic projection
    (network.py:497-514)."""
    th_bt, logs_t = z[:, :58], z[:, -3:]
    out = mano_ref.wrapper_forward(tb, th_bt[:, :48], th_bt[:, -10:])
    xyz, root, s = mano_ref.normalize_pose3d(out["mano_joints"], ROOT_IDX, NORM_IDX)
    verts = (out["mesh"] - root) / s[:, None, None]
    uv = mano_ref.orth_proj(xyz, torch.exp(logs_t[:, 0:1]), logs_t[:, 1:3], image_size, inv_norm)
    return {"xyz": xyz, "verts": verts, "uv": uv}


def forward_log_p(tb, z, y, N):
    """_forward_log_p, network.py:612-667 with mods=['uv'], use_gt=[], T=1."""
    dec = decode(tb, z)
    mu = dec["uv"].flatten(-2)
    w = y["vis"][..., None].repeat(N, 1, 2).flatten(-2)
    out = {"log_p_uv_giv_z": laplace_log_prob(y["crop_uv"].repeat(N, 1), mu, w)}
    th3, th45, bt = z[:, :3], z[:, 3:48], z[:, 48:58]
    out["log_p_th3"] = approx_uniform_ball(th3, math.pi, 5.0)          # network.py:433-434
    out["log_p_th45"] = approx_uniform_rec(th45, -2.0, 2.0, TH45_ALPHA)  # network.py:429
    out["log_p_bt"] = approx_uniform_rec(bt, -0.03, 0.03, 50.0)        # network.py:435
    out["log_p"] = out["log_p_uv_giv_z"] + out["log_p_th3"] + out["log_p_th45"] + out["log_p_bt"]
    return out


def reverse_kld(sd, tb, feat, y, z0, N):
    """_reverse_kld, network.py:760-831, from the conditioning feature on."""
    z = sample_q(sd, feat, z0, N)
    out = {"th_norm": z[:, :48].norm(p=2, dim=1), "bt_norm": z[:, 48:58].norm(p=2, dim=1)}
    lp = forward_log_p(tb, z, y, N)
    out["q_log_p_z_giv_y"] = lp["log_p"].reshape(N, -1).mean(0)
    log_q = flows_ref.log_prob(sub(sd, "q_z_giv_i."), z[:, 3:48], feat.repeat(N, 1))   # network.py:669-701,801
    out["h_q_z_giv_i"] = (-log_q).reshape(N, -1).mean(0)
    out["log_p"] = out["h_q_z_giv_i"] + out["q_log_p_z_giv_y"]
    out["_z"] = z
    out["_log_q"] = log_q
    return out


def get_loss(sd, tb, x, y, z0, N, arch="resnet50", training=True, bf16_storage=False):
    """MHEnt.get_loss, network.py:838-844."""
    feat = encoder(sd, x, arch, training, bf16_storage)
    return reverse_kld(sd, tb, feat, y, z0, N)


def sample(sd, tb, feat, z0, N, image_size=256, N_quant=None):
    """MHEnt.sample, network.py:846-883 with mods {'uv','xyz','verts'}; z0 already times temp.
    N_quant < N keeps, per image, the N_quant hypotheses of highest log q (network.py:866-871)."""
    B = feat.shape[0]
    z = sample_q(sd, feat, z0, N).reshape(N, B, -1)
    if N_quant is not None and N_quant < N:
        log_q = flows_ref.log_prob(sub(sd, "q_z_giv_i."), z.flatten(0, 1)[:, 3:48], feat.repeat(N, 1)).reshape(N, -1)
        idx = torch.topk(log_q, N_quant, dim=0)[1]
        z = torch.gather(z, 0, idx[..., None].repeat(1, 1, z.shape[-1]))
        N = N_quant
    out = {"th_bt": z[..., :58], "logs_t": z[..., -3:]}
    dec = decode(tb, z.reshape(N * B, -1), image_size, inv_norm=True)
    for k in ("verts", "xyz", "uv"):
        out[k] = dec[k].reshape(N, B, -1)
    out["faces"] = tb["th_faces"]
    return out
