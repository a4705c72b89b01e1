"""MHEnt - image encoder -> K pose/shape hypotheses with log-probability -> MANO
joints -> entropy + 2D re-projection ELBO - with the reference's module API
(reference hand/network.py: `BasicEnc` :27-140, `MHEnt` :309-887) on HIP kernels.

Kept from the reference: constructor (`MHEnt(special_cfg, **common_cfg)`), child
module names (feat_extractor, q_z_giv_i, mano_dec, det_head -> identical
state_dict keys), `get_loss` / `log_prob` / `sample` / `training_step_start`
signatures and the keys + shapes of the dicts they return.
Extensions (all optional, defaults reproduce the reference):
  * `N=` hypotheses per image for the loss (the reference hard-codes 10, :780),
  * `noise=` host-supplied base noise for reproducible parity,
  * `fused_entropy=True`: log q from the sampling pass instead of a second,
    inverse pass through the flow (same value to fp32 round-off, SURVEY.md A2 ii);
    set False to run the reference's two-pass form.
The `q_z_giv_i_model='glow'` branch (hand/network.py:342-344,736-742) runs on mhentropy_amd/glow.py's ConditionalGlow - a
restatement of the published nflows algorithm, parity UNPINNED (the third-party class is absent from the reference tree).
Dead reference branches (VAE prior, renderer, GT evidences) raise NotImplementedError.
"""
from typing import Union

import numpy as np
import torch
from torch import nn

from . import ops, resnet
from .ManoLayer import ManoLayer
from .flows import RealNVP
from .glow import ConditionalGlow


class BasicEnc(nn.Module):
    """reference hand/network.py:27-140: ResNet trunk + two Linear heads; returns (z, mn, sd) with sd = exp(l2 / 2) (or
    sigmoid, `sigma_act`) and z = mn + sd * eps, eps ~ N(0, I) (z = mn when deterministic) - `full_outputs=True`, the default
    of the stand-alone class.  MHEnt consumes only `mn` (:779,862); l2 / exp / the epsilon draw are dead for it, so it
    builds its encoder with `full_outputs=False` (-> (mn, mn, None), no l2 product, no draw).  `eps=` supplies the noise for
    reproducible runs; otherwise it is drawn on the device (ops.randn) - a device generator cannot follow the reference's CPU stream."""
    def __init__(self, cfg=None, n_latent: Union[int, list] = 64, backbone="resnet18", pretrained=True,
                 conditional_p=False, K=21, D=3, feat_dim=None, sigma_act="exp", deterministic=False,
                 compute_dtype=torch.float32, full_outputs=True, **kwargs):
        super().__init__()
        if conditional_p:
            raise NotImplementedError("conditional_p is deprecated in the reference (network.py:63)")
        self.n_latent = [n_latent, n_latent] if isinstance(n_latent, int) else list(n_latent)
        if backbone not in resnet.CFG:
            raise NotImplementedError(backbone)
        self.res = resnet.ResNetTrunk(backbone, compute_dtype=compute_dtype)      # random init: no network for pretrained weights
        feat_dim = feat_dim or resnet.CFG[backbone][2]
        self.l1 = nn.Sequential(nn.Linear(feat_dim, self.n_latent[0]))
        self.l2 = nn.Sequential(nn.Linear(feat_dim, self.n_latent[1]))
        if sigma_act not in ("exp", "sigmoid"):
            raise NotImplementedError(f"sigma_act={sigma_act!r}")
        self.sigma_act, self.deterministic, self.full_outputs = sigma_act, deterministic, full_outputs
        self._feat = None

    def forward(self, x, deterministic=False, p=None, eps=None):
        f = self.res(x)
        self._feat = f
        # bf16 mode: the same launch also leaves mn as the bf16 operand of the flow's conditioning product (read by RealNVP._cond_table)
        mn = ops.linear(f, self.l1[0].weight.detach(), self.l1[0].bias.detach(), want_bf16=self.res.compute_dtype == torch.bfloat16)
        if isinstance(mn, tuple):
            mn, mnb = mn
            if mnb is not None:
                mn._mhe_bf16 = mnb
        if not self.full_outputs:
            return mn, mn, None
        l2 = ops.linear(f, self.l2[0].weight.detach(), self.l2[0].bias.detach())
        if mn.shape != l2.shape:
            # list-valued n_latent: the reference returns z = mn with sd in l2's OWN shape (hand/network.py:133-136); the fused launch
            # wants one shape, so sd comes from a deterministic launch on l2 alone (its z output, a copy of l2, is dropped)
            sd, _ = ops.reparam(l2, l2, None, sigmoid_act=self.sigma_act == "sigmoid", deterministic=True)
            return mn, mn, sd
        det = bool(self.deterministic or deterministic)      # hand/network.py:133-136
        if not det and eps is None:
            eps = ops.randn(mn.shape[0], mn.shape[1], mn.device)
        sd, z = ops.reparam(mn, l2, None if det else eps.contiguous(), sigmoid_act=self.sigma_act == "sigmoid", deterministic=det)
        return z, mn, sd


class MHEnt(nn.Module):
    def __init__(self, special_cfg, **common_cfg):
        super().__init__()
        self.integrated = True
        if common_cfg["input"] != "image":
            raise NotImplementedError
        self.feat_extractor = BasicEnc(**dict(common_cfg, full_outputs=False))      # only mn is consumed (hand/network.py:779,862)
        model = special_cfg["q_z_giv_i_model"]
        if model == "realnvp":
            self.q_z_giv_i = RealNVP(**special_cfg["q_z_giv_i_cfg"])
        elif model == "glow":       # reference hand/network.py:342-344; parity unpinned (third-party class absent, see glow.py)
            self.q_z_giv_i = ConditionalGlow(45, 512, 4, 2, context_features=512, dropout_probability=0.2)
        else:
            raise NotImplementedError(f"q_z_giv_i_model={model!r}")
        self.ds = special_cfg["ds"]
        if self.ds not in ("rhd", "ho3d"):
            raise NotImplementedError(self.ds)
        self.image_size = max(special_cfg["image_size"])
        mano_cfg = special_cfg["mano_cfg"]
        self.mano_dec = ManoLayer(skeidx="RHD", flat_hand_mean=mano_cfg["flat_hand_mean"], ncomps=mano_cfg["ncomps"],
                                  use_pca=mano_cfg["use_pca"], output_size=self.image_size, mask_sz=64,
                                  tables=mano_cfg.get("tables"),
                                  **({"MANO_dir": mano_cfg["MANO_dir"]} if "MANO_dir" in mano_cfg else {}))
        self.zdims = {"th3": 3, "th45": 45, "bt": 10, "logs": 1, "t": 2}
        self.zdets = {"th3": True, "th45": False, "bt": True, "logs": True, "t": True}
        self.z_dim = 45
        feat_dim = 512
        self.det_head = nn.Sequential(nn.Linear(feat_dim, feat_dim), nn.ReLU(inplace=True), nn.Linear(feat_dim, 16))
        self.b_2d = float(special_cfg["data_prior_cfg"]["b_2d"])                   # network.py:392
        prior_cfg = special_cfg["prior_cfg"]
        if prior_cfg.get("p_theta45_pth"):
            raise NotImplementedError("VAE4Pose prior is undefined in the reference (network.py:423)")
        self.th45_ref_alpha = float(prior_cfg.get("th45_ref_alpha", 50.0))       # network.py:427
        self.kld_w = self.kld_w_final = special_cfg.get("kld_w", 1.0)
        self.kld_w_annealing = special_cfg["kld_w_annealing"]
        self.T = special_cfg.get("T", 1.0)
        if self.T != 1.0:
            raise NotImplementedError("T != 1")
        self.entropy = special_cfg["loss_cfg"]["entropy"]
        self.best_mode = special_cfg["loss_cfg"]["mode"]
        self._extra_ws = {"log_p_vis_giv_z": 1.0}
        self.loss_N = 10                        # network.py:780
        self.fused_entropy = True

    # ---- pieces ---------------------------------------------------------------
    def _det(self, feat):
        """det_head (network.py:380-383) on the B image rows."""
        h = ops.linear(feat, self.det_head[0].weight.detach(), self.det_head[0].bias.detach(), relu=True)
        return ops.linear(h, self.det_head[2].weight.detach(), self.det_head[2].bias.detach())

    def _noise(self, rows, temp, noise, device):
        """z0 = prior.sample((N*B,)) * temp (hand/flows.py:339, hand/network.py:733-735): drawn on the device inside the step unless the
        caller supplies `noise` (parity runs: a device generator cannot reproduce the reference's CPU stream)"""
        if noise is None:
            return ops.randn(rows, 45, device, scale=temp)
        noise = noise.reshape(rows, 45)
        return (noise * temp).contiguous() if temp != 1.0 else noise.contiguous()

    def _glow_sample(self, feat, N, temp, noise):
        """reference hand/network.py:736-742: noise (B,N,45)*temp -> sample_and_log_prob -> rows permuted to sample-major.
        Here the flow runs directly on sample-major rows (row n*B+b uses feat[b]); a `noise` given in the reference's
        (B,N,45) layout is re-ordered, a (N*B,45) one is taken as sample-major."""
        B = feat.shape[0]
        if noise is not None and noise.dim() == 3:
            noise = noise.permute(1, 0, 2).reshape(N * B, 45)
        z0 = self._noise(N * B, temp, noise, feat.device)
        return self.q_z_giv_i._run(z0, feat.contiguous(), True, 1, B)

    # ---- reference surface ------------------------------------------------------
    def _reverse_kld(self, y, x, mods=None, return_dict=True, N=None, noise=None):
        """reference hand/network.py:760-831."""
        if mods is not None and list(mods) != ["uv"]:
            raise NotImplementedError("only the weakly supervised 'uv' likelihood of the shipped config is built")
        N = N or self.loss_N
        tr = getattr(self, "_trainer", None)
        if tr is not None and self.training and torch.is_grad_enabled():
            # a train.TrainStep is attached: the loss dict comes out as ONE autograd node whose backward is the
            # hand-written reverse pass, so the reference's `total_loss.backward()` works unchanged
            from .train import differentiable_get_loss
            return differentiable_get_loss(tr, x, y, N=N, noise=noise)
        _, feat, _ = self.feat_extractor(x)
        B = feat.shape[0]
        if isinstance(self.q_z_giv_i, ConditionalGlow):      # entropy from the sampling pass itself (network.py:781-783,798-799)
            th45, log_q = self._glow_sample(feat, N, 1.0, noise)
        elif self.entropy and self.fused_entropy:
            z0 = self._noise(N * B, 1.0, noise, feat.device)
            th45, log_q = self.q_z_giv_i.sample_with_log_prob(z0, feat)
        else:
            z0 = self._noise(N * B, 1.0, noise, feat.device)
            th45 = self.q_z_giv_i.forward_p(z0, cond=feat)
            log_q = self.q_z_giv_i.log_prob(th45, logvar=feat) if self.entropy else None       # network.py:801
        o = ops.mano_joints(th45, self._det(feat), self.mano_dec.table_blob(), y["crop_uv"].contiguous(),
                            y["vis"].contiguous(), self.b_2d, self.th45_ref_alpha, want=("log_p", "norms"))
        q_log_p, h, log_p = ops.elbo_reduce(o["log_p"], log_q, N, B)
        out = {"th_norm": o["norms"][:, 0], "bt_norm": o["norms"][:, 1], "q_log_p_z_giv_y": q_log_p}
        if self.entropy:
            out["h_q_z_giv_i"] = h
            out["log_p"] = log_p
        else:
            out["log_p"] = q_log_p
        if not return_dict:
            raise NotImplementedError
        return out

    def log_prob(self, y, x, div_type=0, mods=None, return_dict=True, **kw):
        return self._reverse_kld(y, x, mods=mods, return_dict=return_dict, **kw)

    def get_loss(self, x, y, div_type=0, mods=None, return_dict=True, **kw):
        """reference hand/network.py:838-844."""
        return self.log_prob(y, x, div_type=div_type, mods=mods, return_dict=return_dict, **kw)

    def sample(self, x, N: Union[int, list] = 5, temp=0.5, mods=None, y=None, noise=None, feat=None):
        """reference hand/network.py:846-883 -> th_bt (N,B,58), logs_t (N,B,3), verts (N,B,2334),
        xyz (N,B,63), uv (N,B,42) in pixels, faces."""
        N_quant = N
        if isinstance(N, (list, tuple)):
            N, N_quant = N
        out = {}
        if y is not None and "image" in y:
            out["image"] = y["image"]
        if feat is None:       # `feat=` (extension): the conditioning feature of a forward already run on x
            _, feat, _ = self.feat_extractor(x)
        B = feat.shape[0]
        if isinstance(self.q_z_giv_i, ConditionalGlow):
            th45, log_q = self._glow_sample(feat, N, temp, noise)
        else:                   # log q comes from the sampling pass itself (only needed for the top-k selection)
            z0 = self._noise(N * B, temp, noise, feat.device)
            if N_quant < N:
                th45, log_q = self.q_z_giv_i.sample_with_log_prob(z0, feat)
            else:
                th45 = self.q_z_giv_i.forward_p(z0, cond=feat)
        if N_quant < N:         # keep the N_quant most likely hypotheses per image (network.py:866-871)
            _, th45 = ops.topk_gather(log_q, th45, N, B, N_quant)
            N = N_quant
        mods = {"xyz", "uv", "verts"} if mods is None else set(mods)
        blob = self.mano_dec.table_blob()
        o = ops.mano_joints(th45, self._det(feat), blob, inv_norm=True, image_size=float(self.image_size),
                            want=("z", "xyz", "uv") + (("verts",) if "verts" in mods else ()))      # the mesh from the joint pass' operands
        z = o["z"].view(N, B, 61)
        out["th_bt"], out["logs_t"] = z[..., :58], z[..., -3:]
        if "verts" in mods:
            out["verts"] = o["verts"].view(N, B, -1)
            out["faces"] = self.mano_dec.mano_faces
        if "xyz" in mods:
            out["xyz"] = o["xyz"].view(N, B, -1)
        if "uv" in mods:
            out["uv"] = o["uv"].view(N, B, -1)
        return out

    def training_step_start(self, step):
        """reference hand/network.py:885-887."""
        kld_w_init, kld_w_steps = self.kld_w_annealing
        self.kld_w = kld_w_init + (self.kld_w_final - kld_w_init) * min(1.0, step / kld_w_steps)
