"""ORACLE tooling (runs ONLY in the build container, where /root/reference exists).

Imports the reference's own Python modules on CPU (recipe: SURVEY.md appendix A6),
feeds them the seeded synthetic weights / tables / inputs of mhentropy_amd.synth,
checks the oracle restatement (oracle/*_ref.py) against them, and writes the
golden input/output vectors to tests/golden/*.npz.  The fixtures hold DATA only
(inputs, seeds, expected outputs); the reference's source never leaves
/root/reference.

    python -m oracle.gen_golden            # regenerate + validate everything

Third-party modules the reference imports but the hot path never executes (cv2,
trimesh, pycocotools, nflows, torchvision, the MANO pickle loader) are replaced by
empty placeholder modules so that `import network` succeeds; none of them
contributes arithmetic to any fixture.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/hand"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from mhentropy_amd import synth  # noqa: E402
from oracle import flows_ref, mano_ref, network_ref, criteria_ref  # noqa: E402


class _Rv:
    """chumpy-like holder: the reference reads `.r` (manolayer.py:70-84)."""
    def __init__(self, a):
        self.r = a


def _install_placeholders(tables):
    import scipy.sparse
    sys.dont_write_bytecode = True
    for name in ("cv2", "trimesh", "pycocotools", "pycocotools.coco", "pycocotools.cocoeval",
                 "nflows", "nflows.flows", "torchvision", "torchvision.models",
                 "mano", "mano.webuser", "mano.webuser.smpl_handpca_wrapper_HAND_only"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pycocotools.coco"].COCO = object
    sys.modules["pycocotools.cocoeval"].COCOeval = object

    class _Glow(torch.nn.Module):
        pass
    sys.modules["nflows.flows"].ConditionalGlow = _Glow

    class _Trunk(torch.nn.Module):
        """stand-in trunk: returns a preset (B,feat) tensor; the encoder's convolutions
        are torchvision's, not the reference's, and are pinned separately."""
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Identity()
            self.fixed = None

        def forward(self, x):
            return self.fc(self.fixed)
    tv = sys.modules["torchvision.models"]
    tv.resnet18 = lambda pretrained=False: _Trunk()
    tv.resnet50 = lambda pretrained=False: _Trunk()
    sys.modules["torchvision"].models = tv

    def ready_arguments(path):
        return {
            "hands_components": tables["hands_components"],
            "hands_mean": tables["hands_mean"],
            "betas": _Rv(tables["betas"]),
            "shapedirs": _Rv(tables["shapedirs"]),
            "posedirs": _Rv(tables["posedirs"]),
            "v_template": _Rv(tables["v_template"]),
            "J_regressor": scipy.sparse.csc_matrix(tables["J_regressor"]),
            "weights": _Rv(tables["weights"]),
            "f": tables["faces"],
            "kintree_table": tables["kintree_table"],
        }
    sys.modules["mano.webuser.smpl_handpca_wrapper_HAND_only"].ready_arguments = ready_arguments

    # hard-coded device='cuda' in the reference (SURVEY.md "Facts") -> cpu
    def _cpu(fn):
        def w(*a, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return w
    for n in ("zeros", "ones", "rand", "randn", "tensor", "empty"):
        setattr(torch, n, _cpu(getattr(torch, n)))
    torch.Tensor.cuda = lambda self, *a, **k: self


def _t(sd):
    return {k: torch.as_tensor(v) for k, v in sd.items()}


def _np(d):
    return {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def _close(name, a, b, rtol=2e-5, atol=2e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-30
    ok = err <= atol + rtol * scale
    print(f"  {'ok ' if ok else 'BAD'} {name:34s} max|diff|={err:.3e} (scale {scale:.3e})")
    assert ok, name


class _FixedPrior:
    """hands the captured base noise to RealNVP.sample (flows.py:339)."""
    def __init__(self, real, noise):
        self._real, self._noise = real, noise

    def sample(self, shape):
        assert tuple(shape) == (self._noise.shape[0],)
        return self._noise.clone()

    def __getattr__(self, k):
        return getattr(self._real, k)


def gen_flow(flows, tag, seed, h, steps, R, cond_dim):
    print(f"[flow_{tag}]")
    sdn = synth.flow_state(seed, 45, cond_dim, (h, h), steps)
    ref = flows.RealNVP(dim=45, tsfm_on=cond_dim, kemb=False, jointN=21, h_dims=[h, h], num_steps=steps)
    ref.load_state_dict(_t(sdn))
    sd = _t(sdn)
    rng = np.random.default_rng(seed + 77)
    z0 = torch.as_tensor(rng.normal(0, 1, (R, 45)).astype(np.float32))
    feat = torch.as_tensor(rng.normal(0, 1, (R, cond_dim)).astype(np.float32))
    with torch.no_grad():
        x_ref = ref.forward_p(z0, cond=ref.make_cond(feat))
        zb_ref, ld_ref = ref.backward_p(x_ref, cond=feat)
        lp_ref = ref.log_prob(x_ref, logvar=feat)
        x, ldf = flows_ref.forward_p_logdet(sd, z0, feat)
        zb, ld = flows_ref.backward_p(sd, x_ref, feat)
        lp = flows_ref.log_prob(sd, x_ref, feat)
    _close("forward_p", x, x_ref)
    _close("backward_p.z", zb, zb_ref)
    _close("backward_p.log_det", ld, ld_ref)
    _close("log_prob", lp, lp_ref)
    _close("fused log q (A2 ii)", flows_ref.std_normal_logprob(z0) - ldf, lp_ref, 1e-4, 1e-4)
    np.savez_compressed(os.path.join(GOLD, f"flow_{tag}.npz"),
                        seed=seed, h=h, steps=steps, cond_dim=cond_dim,
                        z0=z0.numpy(), feat=feat.numpy(), x=x_ref.numpy(), z_back=zb_ref.numpy(),
                        log_det=ld_ref.numpy(), log_prob=lp_ref.numpy())


def gen_mano(ManoWrapper, tables):
    print("[mano]")
    tb = mano_ref.tables_from_numpy(tables)
    layer = ManoWrapper(skeidx="RHD", flat_hand_mean=False, ncomps=45, use_pca=True, output_size=256, mask_sz=64)
    rng = np.random.default_rng(5)
    R = 6
    theta = rng.normal(0, 0.6, (R, 48)).astype(np.float32)
    beta = rng.normal(0, 0.02, (R, 10)).astype(np.float32)
    theta[0] = 0.0                      # zero pose (Rodrigues at the +1e-8 floor)
    theta[1, :3] = 1e-9                 # near-zero root rotation
    theta[2, :3] = [3.0, 0.5, -0.2]     # large root rotation
    beta[0] = 0.0
    theta, beta = torch.as_tensor(theta), torch.as_tensor(beta)
    with torch.no_grad():
        ref = layer(beta=beta, theta=theta)
        out = mano_ref.wrapper_forward(tb, theta, beta)
    for k in ("mesh", "mano_joints", "joints"):
        _close(k, out[k], ref[k], 2e-5, 2e-4)   # mm units, |x| ~ 1e2
    np.savez_compressed(os.path.join(GOLD, "mano.npz"), table_seed=0, theta=theta.numpy(), beta=beta.numpy(),
                        mesh=ref["mesh"].numpy(), mano_joints=ref["mano_joints"].numpy(), joints=ref["joints"].numpy())


def _mhent(network, tables, seed, h, steps):
    special = dict(
        q_z_giv_i_model="realnvp",
        q_z_giv_i_cfg=dict(dim=45, tsfm_on=512, kemb=False, jointN=21, h_dims=[h, h], num_steps=steps),
        ds="ho3d", image_size=[256, 256],
        mano_cfg=dict(flat_hand_mean=False, ncomps=45, use_pca=True),
        prior_cfg=dict(p_theta45_pth=None, th45_ref_alpha=50),
        data_prior_cfg=dict(b_2d=0.03, w_prior_2d=0),
        loss_cfg=dict(entropy=True, mode=False, w_reg_ds=0),
        kld_w=1, kld_w_annealing=[1, 24000], T=1.0)
    common = dict(n_latent=512, backbone="resnet50", pretrained=False, conditional_p=False, K=21, D=3,
                  feat_dim=None, sigma_act="exp", deterministic=False, input="image")
    model = network.MHEnt(special, **common)
    sdn = {}
    sdn.update({"q_z_giv_i." + k: v for k, v in synth.flow_state(seed, 45, 512, (h, h), steps).items()})
    sdn.update(synth.head_state(seed, 2048, 512, 16))
    missing, unexpected = model.load_state_dict(_t(sdn), strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("mano_dec.") for k in missing), missing
    return model, sdn


def gen_mhent(network, criteria, tables, tag, seed, h, steps, B, Ns):
    print(f"[mhent_{tag}]")
    model, sdn = _mhent(network, tables, seed, h, steps)
    model.train()
    sd = _t(sdn)
    tb = mano_ref.tables_from_numpy(tables)
    _, yn = synth.batch(seed, B, with_image=False)
    y = _t(yn)
    y["image"] = torch.zeros(B, 1)
    rng = np.random.default_rng(seed + 99)
    trunk = torch.as_tensor(rng.normal(0, 0.5, (B, 2048)).astype(np.float32))
    model.feat_extractor.res.fixed = trunk
    x_dummy = torch.zeros(B, 3, 8, 8)

    # ---- get_loss: the reference hard-codes N = 10 (network.py:780)
    N = 10
    z0 = torch.as_tensor(synth.noise(seed, N * B))
    real_prior = model.q_z_giv_i.prior
    model.q_z_giv_i.prior = _FixedPrior(real_prior, z0)
    ref = model.get_loss(x_dummy, y, mods=["uv"])
    feat = torch.nn.functional.linear(trunk, sd["feat_extractor.l1.0.weight"], sd["feat_extractor.l1.0.bias"])
    feat_req = feat.clone().requires_grad_(True)
    sd_g = dict(sd)
    gnames = ["det_head.2.weight", "q_z_giv_i.s.0.l.0.weight", f"q_z_giv_i.t.{2 * steps - 1}.l.2.weight"]
    for n in gnames:
        sd_g[n] = sd[n].clone().requires_grad_(True)
    out = network_ref.reverse_kld(sd_g, tb, feat_req, y, z0, N)
    for k in ("th_norm", "bt_norm", "q_log_p_z_giv_y", "h_q_z_giv_i", "log_p"):
        _close("get_loss." + k, out[k], ref[k], 5e-5, 5e-4)
    # gradients of the training loss (criteria.py:55,173) for later backward-kernel parity
    loss_ref = (-ref["log_p"]).mean()
    pr = dict(model.named_parameters())
    g_ref = torch.autograd.grad(loss_ref, [pr[n] for n in gnames] + [pr["feat_extractor.l1.0.bias"]])
    g = torch.autograd.grad((-out["log_p"]).mean(), [sd_g[n] for n in gnames] + [feat_req])
    for n, a, b in zip(gnames, g[:3], g_ref[:3]):
        _close("grad " + n, a, b, 2e-4, 1e-6)
    _close("grad feat (sum over B == dL/d l1.bias)", g[3].sum(0), g_ref[3], 2e-4, 1e-6)
    # per-term log-likelihood / prior values through the reference's own helper
    with torch.no_grad():
        terms_ref = model._forward_log_p(out["_z"].detach(), y, use_gt=[], mods=["uv"], feat=feat)
        terms = network_ref.forward_log_p(tb, out["_z"].detach(), y, N)
    for k in ("log_p_uv_giv_z", "log_p_th3", "log_p_th45", "log_p_bt", "log_p"):
        _close("terms." + k, terms[k], terms_ref[k], 5e-5, 5e-4)
    gold = dict(seed=seed, h=h, steps=steps, B=B, N_loss=N, trunk=trunk.numpy(), z0_loss=z0.numpy(),
                feat=feat.numpy(), z_loss=out["_z"].detach().numpy(), log_q_loss=out["_log_q"].detach().numpy())
    gold.update({"y_" + k: v for k, v in yn.items()})
    gold.update({"loss_" + k: ref[k].detach().numpy() for k in ("th_norm", "bt_norm", "q_log_p_z_giv_y", "h_q_z_giv_i", "log_p")})
    gold.update({"terms_" + k: terms_ref[k].numpy() for k in ("log_p_uv_giv_z", "log_p_th3", "log_p_th45", "log_p_bt")})
    gold.update({"grad_" + n: a.numpy() for n, a in zip(gnames, g_ref[:3])})
    gold["grad_feat"] = g[3].detach().numpy()

    # ---- other hypothesis counts: drive the reference through the helpers that take N
    for Nk in Ns:
        z0k = torch.as_tensor(synth.noise(seed + Nk, Nk * B))
        model.q_z_giv_i.prior = _FixedPrior(real_prior, z0k)
        with torch.no_grad():
            zk = model._sample_q_z_giv_i(feat, N=Nk, y=y)
            lpk = model._forward_log_p(zk, y, use_gt=[], mods=["uv"], feat=feat)["log_p"]
            lqk = model._reverse_log_q(zk, feat.repeat(Nk, 1))
            outk = network_ref.reverse_kld(sd, tb, feat, y, z0k, Nk)
        _close(f"N={Nk} z", outk["_z"], zk)
        _close(f"N={Nk} log_p rows", network_ref.forward_log_p(tb, zk, y, Nk)["log_p"], lpk, 5e-5, 5e-4)
        _close(f"N={Nk} log_q", outk["_log_q"], lqk, 5e-5, 5e-4)
        gold[f"z0_N{Nk}"] = z0k.numpy()
        gold[f"z_N{Nk}"] = zk.numpy()
        gold[f"logp_rows_N{Nk}"] = lpk.numpy()
        gold[f"logq_N{Nk}"] = lqk.numpy()

    # ---- sample() (network.py:846-883) at N = 4, temp 0.8 (CrossModalHand.py:360)
    Ns_ = 4
    z0s = torch.as_tensor(synth.noise(seed + 500, Ns_ * B)) * 0.8
    model.q_z_giv_i.prior = _FixedPrior(real_prior, z0s / 0.8)
    with torch.no_grad():
        sref = model.sample(x_dummy, N=[Ns_, Ns_], temp=0.8, mods={"uv", "xyz", "verts"}, y=y)
        s = network_ref.sample(sd, tb, feat, z0s, Ns_)
    for k in ("th_bt", "logs_t", "verts", "xyz", "uv"):
        _close("sample." + k, s[k], sref[k], 5e-5, 5e-4)
    gold["z0_sample"] = z0s.numpy()
    gold.update({"sample_" + k: sref[k].numpy() for k in ("th_bt", "logs_t", "verts", "xyz", "uv")})

    # ---- sample() with top-k hypothesis selection (network.py:866-871): keep the N_quant most likely of N
    Nt, Qt = 6, 3
    z0t = torch.as_tensor(synth.noise(seed + 700, Nt * B)) * 0.8
    model.q_z_giv_i.prior = _FixedPrior(real_prior, z0t / 0.8)
    with torch.no_grad():
        tref = model.sample(x_dummy, N=[Nt, Qt], temp=0.8, mods={"uv", "xyz", "verts"}, y=y)
        tk = network_ref.sample(sd, tb, feat, z0t, Nt, N_quant=Qt)
    for k in ("th_bt", "logs_t", "verts", "xyz", "uv"):
        _close("sample_topk." + k, tk[k], tref[k], 5e-5, 5e-4)
    gold["z0_topk"] = z0t.numpy()
    gold.update({"topk_" + k: tref[k].numpy() for k in ("th_bt", "logs_t", "verts", "xyz", "uv")})

    # ---- criterion + metrics (criteria.py:47-173)
    crit = criteria.MHEntLoss()
    o = {"log_p": ref["log_p"].detach(), "xyz": sref["xyz"], "uv": sref["uv"], "verts": sref["verts"]}
    with torch.no_grad():
        tot_ref, losses_ref, met_ref = crit(dict(o), y)
        tot, losses, met = criteria_ref.mhent_loss(dict(o), y)
    _close("criterion.total", tot, tot_ref)
    assert set(met) == set(met_ref), (sorted(met), sorted(met_ref))
    for k in sorted(met_ref):
        _close("metric." + k, met[k], met_ref[k], 5e-5, 5e-5)
        gold["metric_" + k] = met_ref[k].numpy()
    gold["criterion_total"] = tot_ref.numpy()
    np.savez_compressed(os.path.join(GOLD, f"mhent_{tag}.npz"), **gold)


def gen_priors(network, tables):
    """the soft-support priors far OUTSIDE their supports (network.py:155-165, 429-435): th3 up to 3 pi (ball of radius pi),
    th45 up to +-5 (box +-2), bt up to +-0.2 (box +-0.03) - the get_loss fixtures sit inside them, where those terms are 0.
    det-head quantities (th3, bt, logs, t) are per image and repeated over the N hypothesis rows, as in the product."""
    print("[priors_outside]")
    model, _ = _mhent(network, tables, 21, 64, 2)
    tb = mano_ref.tables_from_numpy(tables)
    B, N = 6, 4
    _, yn = synth.batch(31, B, with_image=False)
    y = _t(yn)
    rng = np.random.default_rng(123)
    det = np.zeros((B, 16), np.float32)                       # th3 | bt | logs | t
    dirs = rng.normal(0, 1, (B, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    det[:, :3] = dirs * (np.pi * np.array([0.5, 0.999, 1.001, 1.5, 2.2, 3.0]))[:, None]
    det[:, 3:13] = rng.uniform(-1, 1, (B, 10)) * np.array([0.02, 0.03, 0.031, 0.06, 0.12, 0.2])[:, None]
    det[:, 13] = rng.normal(0, 0.3, B); det[:, 14:16] = rng.normal(0, 0.2, (B, 2))
    th45 = (rng.uniform(-1, 1, (N * B, 45)) * rng.choice([0.5, 1.9, 2.0, 2.1, 3.0, 5.0], (N * B, 1))).astype(np.float32)
    th45[0, :] = 2.0; th45[1, :] = -2.0                         # exactly on the box faces
    d = np.tile(det, (N, 1))
    z = torch.as_tensor(np.concatenate([d[:, :3], th45, d[:, 3:13], d[:, 13:14], d[:, 14:16]], 1).astype(np.float32))
    with torch.no_grad():
        ref = model._forward_log_p(z, y, use_gt=[], mods=["uv"], feat=torch.zeros(B, 512))
        mine = network_ref.forward_log_p(tb, z, y, N)
    keys = ("log_p_uv_giv_z", "log_p_th3", "log_p_th45", "log_p_bt", "log_p")
    for k in keys:
        _close("terms." + k, mine[k], ref[k], 5e-5, 5e-4)
    assert float(ref["log_p_th3"].min()) < -10 and float(ref["log_p_bt"].min()) < -100 and float(ref["log_p_th45"].min()) < -1000
    gold = {"B": B, "N": N, "z": z.numpy(), "det": det, "th45": th45}
    gold.update({"y_" + k: v for k, v in yn.items()})
    gold.update({"terms_" + k: ref[k].numpy() for k in keys})
    np.savez_compressed(os.path.join(GOLD, "priors_outside.npz"), **gold)


def gen_rot6d():
    """6D rotation representation -> R (manopth/rot6d.py:4-24 and the 'robust' variant :26-51), the first piece of the body-model
    path (reached at manopth/manolayer.py:150-156)"""
    print("[rot6d]")
    from manopth import rot6d        # reference module
    from oracle import rot6d_ref
    rng = np.random.default_rng(66)
    poses = rng.normal(0, 1, (64, 6)).astype(np.float32)
    poses[0] = [1, 0, 0, 0, 1, 0]                    # already orthonormal
    poses[1] = [2, 0, 0, 1, 1, 0]                    # needs Gram-Schmidt
    poses[2] = [1e-3, 0, 0, 0, 2e-3, 0]              # small magnitudes (normalisation)
    poses[3] = [0, 0, 5, 0.3, 0, 5]                  # nearly parallel pair
    pt = torch.as_tensor(poses)
    with torch.no_grad():
        ref = rot6d.compute_rotation_matrix_from_ortho6d(pt)
        rob = rot6d.robust_compute_rotation_matrix_from_ortho6d(pt)
    _close("rot6d", rot6d_ref.rotation_from_ortho6d(pt), ref, 1e-6, 1e-6)
    _close("rot6d robust", rot6d_ref.rotation_from_ortho6d_robust(pt), rob, 1e-6, 1e-6)
    # gradient of a fixed linear functional of R, for the reverse kernel
    w = torch.as_tensor(rng.normal(0, 1, (64, 3, 3)).astype(np.float32))
    pr = pt.clone().requires_grad_(True)
    (rot6d.compute_rotation_matrix_from_ortho6d(pr) * w).sum().backward()
    pm = pt.clone().requires_grad_(True)
    (rot6d_ref.rotation_from_ortho6d(pm) * w).sum().backward()
    _close("rot6d grad", pm.grad, pr.grad, 1e-5, 1e-6)
    np.savez_compressed(os.path.join(GOLD, "rot6d.npz"), poses=poses, R=ref.numpy(), R_robust=rob.numpy(), w=w.numpy(),
                        grad_poses=pr.grad.numpy())


HO3D_CASES = [(0, (0, 0), False), (0, (0, 0), True), (1, (250, -180), False), (1, (250, -180), True), (2, (-280, 200), True),
              (3, (120, 60), True)]


def gen_ho3d():
    """row f4: the reference's `Generate_ho3d_uv.__getitem__` (hand/dataloader/ho3d_dataloader.py:272-459) on synthetic decoded
    samples.  The module is imported from a scratch directory holding the (empty) dataset directories its import-time checks look
    for; its file readers are pointed at the synthetic arrays, and the OpenCV / torchvision calls it makes are served by the
    restatements of oracle/ho3d_ref.py (those primitives stay unpinned; the reference's own logic around them is what is pinned)."""
    import tempfile
    from oracle import ho3d_ref
    scratch = tempfile.mkdtemp(prefix="ho3d_ref_")
    for d in ("datasets/HO3D_v3/HO3D_v3", "datasets/HO3D_v3/models", "datasets/HO3D_v3/HO3D/data"):
        os.makedirs(os.path.join(scratch, d))
    for name in ("imageio", "torchvision.transforms", "torchvision.transforms.functional", "easydict", "yacs", "yacs.config", "tensorboardX",
                 "skimage", "skimage.transform"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tvt = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].transforms = tvt
    tvt.functional = sys.modules["torchvision.transforms.functional"]
    tvt.functional.erase = None
    cwd = os.getcwd()
    os.chdir(scratch)
    try:
        import dataloader.ho3d_dataloader as hd                 # reference module
    finally:
        os.chdir(cwd)
    cv = hd.cv2
    cv.INTER_NEAREST, cv.BORDER_CONSTANT = 0, 0
    cv.resize = lambda img, dsize, interpolation=None: ho3d_ref.resize_nearest(img, dsize)
    cv.Rodrigues = lambda r: (ho3d_ref.rodrigues(r), None)
    cv.copyMakeBorder = lambda img, t, b, l, r, kind, value=None: ho3d_ref.copy_make_border(img, t, b, l, r, value)
    cv.getRotationMatrix2D = ho3d_ref.get_rotation_matrix_2d
    cv.warpAffine = lambda img, M, dsize, flags=None, borderValue=0.0: ho3d_ref.warp_affine_nearest(img, M, dsize)

    class _Compose:
        def __init__(self, transforms): self.transforms = transforms
        def __call__(self, x):
            for t in self.transforms: x = t(x)
            return x
    tvt.Compose = hd.torchvision.transforms.Compose = _Compose
    tvt.ToPILImage = lambda: (lambda a: a)
    tvt.ToTensor = lambda: (lambda a: torch.from_numpy(a.astype(np.float32).transpose(2, 0, 1) / np.float32(255)))
    tvt.Normalize = lambda m, s: (lambda t: (t - torch.tensor(m, dtype=torch.float32)[:, None, None]) / torch.tensor(s, dtype=torch.float32)[:, None, None])
    cur = {}
    hd.imageio.imread = lambda fn: cur["s"]["seg"] if fn.endswith(".png") else cur["s"]["image"]
    hd.read_depth_img = lambda *a: ho3d_ref.decode_depth(cur["s"]["depth_png"])
    hd.read_annotation = lambda *a: {"objName": "obj", "objRot": cur["s"]["obj_rot"], "objTrans": cur["s"]["obj_trans"], "camMat": cur["s"]["cam"]}
    for ci, (seed, off, aug) in enumerate(HO3D_CASES):
        smp = synth.ho3d_sample(seed, off)
        cur["s"] = smp
        ds = object.__new__(hd.Generate_ho3d_uv)
        ds.train_file, ds.baseDir, ds.model = np.array(["SEQ/0000"]), "", "train"
        ds.handJoints3D, ds.handMesh = smp["joints3d"][None], smp["mesh"][None]
        ds.objmesh_all = {"obj": {"v": smp["obj_verts"], "vn": smp["obj_verts"]}}
        ds.dpda, ds.aug, ds.joint_idx = "HO3D", aug, "RHD"
        np.random.seed(100 + seed)
        img, tgt = ds[0]
        prm = synth.ho3d_aug_params(100 + seed) if aug else None
        oimg, ot = ho3d_ref.getitem(smp, prm)
        _close(f"ho3d[{ci}] image", torch.as_tensor(oimg), img, 0, 0)
        for k in ("crop_uv", "vis", "depth", "original_pose3d", "verts", "pose3d", "st", "scale", "crop_center", "crop_size", "pose3d_root",
                  "camera", "rot_mat_inv", "_rot_mat", "uvd"):
            _close(f"ho3d[{ci}] {k}", torch.as_tensor(ot[k]), tgt[k].reshape(ot[k].shape), 1e-6, 1e-6)
        for k in ("hand_mask", "object_mask"):
            assert np.array_equal(ot[k], tgt[k].numpy()), k
        u8 = np.rint((img.numpy() * 0.5 + 0.5) * 255).astype(np.uint8)
        out = {k: np.asarray(tgt[k]) for k in ("crop_uv", "vis", "original_pose3d", "pose3d", "st", "scale", "crop_center", "crop_size", "pose3d_root",
                                               "rot_mat_inv", "_rot_mat", "uvd")}
        out.update(seed=seed, offset=np.asarray(off), aug=int(aug), aug_seed=100 + seed, image_u8_sub=u8[:, ::4, ::4], image_u8_sum=u8.astype(np.int64).sum((1, 2)),
                   hand_mask=np.packbits(tgt["hand_mask"].numpy()), object_mask=np.packbits(tgt["object_mask"].numpy()),
                   depth_sub=tgt["depth"].numpy()[::4, ::4], depth_sum=np.float64(tgt["depth"].numpy().astype(np.float64).sum()),
                   verts_sum=np.float64(tgt["verts"].numpy().astype(np.float64).sum()))
        np.savez_compressed(os.path.join(GOLD, f"ho3d_{ci}.npz"), **out)


def main():
    os.makedirs(GOLD, exist_ok=True)
    tables = synth.mano_tables(0)
    _install_placeholders(tables)
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")
    import flows          # noqa: E402  (reference module)
    import network        # noqa: E402
    import criteria       # noqa: E402
    import ManoLayer as ManoWrapperMod  # noqa: E402
    os.chdir(cwd)
    torch.manual_seed(0)
    gen_flow(flows, "small", 11, 64, 2, 8, 512)
    gen_flow(flows, "shipped", 12, 512, 6, 40, 512)
    gen_mano(ManoWrapperMod.ManoLayer, tables)
    gen_mhent(network, criteria, tables, "small", 21, 64, 2, 2, (4,))
    gen_mhent(network, criteria, tables, "shipped", 22, 512, 6, 3, (16,))
    gen_priors(network, tables)
    gen_rot6d()
    gen_ho3d()
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
