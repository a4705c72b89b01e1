"""Body-model decoder of a ProHMR-style multi-hypothesis head (reference README.md:26-42):
K samples of a 144-D pose (24 joints x 6D rotation) per image -> rotation matrices (reference hand/manopth/rot6d.py:4-24)
-> linear-blend skinning of a 24-joint / 6,890-vertex body (the arithmetic of hand/manopth/manolayer.py:181-246 at SMPL's
sizes), on the HIP kernels of csrc/body.hip.

PARITY: `rot6d` is in the reference tree and pinned by fixtures generated from it (tests/golden/rot6d.npz).  The skinning
arithmetic is pinned at the MANO sizes through oracle/body_ref.py == oracle/mano_ref.py (itself pinned by reference
fixtures).  ProHMR's SMPLFlow / SMPL classes and the SMPL model file are out of tree (README.md:30; licence): at body size
the tables are synthetic and parity with ProHMR itself is UNPINNED.

Hypotheses are independent given the conditioning feature, so `forward` takes any slice of the K hypotheses: the
hypothesis-sharded form is `dist.HypothesisShards` around this layer."""
import ctypes as C

import os

import numpy as np
import torch
from torch import nn

from . import ops, _lib

# SMPL's kinematic tree (24 joints; published with the model, Loper et al. 2015)
SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)


def synthetic_body_tables(seed=0, NV=6890, J=24, nb=10, parents=SMPL_PARENTS):
    """SMPL-SHAPED random tables (the real model is licence-restricted): template ~ N(0, 0.3), small blend shapes,
    positive row-normalised skinning weights concentrated on 4 joints per vertex, a joint regressor with rows summing to 1."""
    rng = np.random.default_rng(seed + 9000)
    f32 = lambda a: np.asarray(a, np.float32)
    w = np.zeros((NV, J))
    for v in range(NV):
        js = rng.choice(J, 4, replace=False)
        w[v, js] = rng.random(4) + 0.05
    w /= w.sum(1, keepdims=True)
    jr = rng.random((J, NV)) * (rng.random((J, NV)) < 0.01)
    jr[:, 0] += 1e-3
    jr /= jr.sum(1, keepdims=True)
    return {"v_template": f32(rng.normal(0, 0.3, (NV, 3))), "shapedirs": f32(rng.normal(0, 0.01, (NV, 3, nb))),
            "posedirs": f32(rng.normal(0, 0.002, (NV, 3, 9 * (J - 1)))), "J_regressor": f32(jr), "weights": f32(w),
            "parents": np.asarray(parents, np.int32)}


def rot6d_to_rotmat(poses6, robust=False):
    """(..., 6) -> (..., 3, 3); reference hand/manopth/rot6d.py:4-24 (robust: :26-51)"""
    p = poses6.reshape(-1, 6)
    ops._chk(p, torch.float32, "rot6d.poses")
    out = torch.empty(p.shape[0], 3, 3, device=p.device, dtype=torch.float32)
    ops.check(_lib.lib().mhe_rot6d_to_rotmat_f32(ops._ptr(p), ops._ptr(out), p.shape[0], int(robust), ops._stream()), "mhe_rot6d_to_rotmat_f32")
    return out.view(*poses6.shape[:-1], 3, 3)


def rot6d_to_rotmat_bwd(poses6, g_rotmats):
    p, g = poses6.reshape(-1, 6), g_rotmats.reshape(-1, 9)
    ops._chk(p, torch.float32, "rot6d.poses"); ops._chk(g, torch.float32, "rot6d.g", (p.shape[0], 9))
    out = torch.empty_like(p)
    ops.check(_lib.lib().mhe_rot6d_to_rotmat_bwd_f32(ops._ptr(p), ops._ptr(g), ops._ptr(out), p.shape[0], ops._stream()), "mhe_rot6d_to_rotmat_bwd_f32")
    return out.view(poses6.shape)


class BodyLayer(nn.Module):
    """linear-blend skinning of a (J, NV) body model from rotation matrices or a 6D pose; buffers carry SMPL's names"""
    def __init__(self, tables):
        super().__init__()
        t = {k: np.asarray(v) for k, v in tables.items()}
        self.NV, self.J, self.nb = t["v_template"].shape[0], t["weights"].shape[1], t["shapedirs"].shape[2]
        if self.J > 32 or t["posedirs"].shape[2] != 9 * (self.J - 1):
            raise ValueError("BodyLayer: J <= 32 and posedirs with 9 (J-1) pose-blend coefficients")
        parents = t["parents"].astype(np.int64)
        if parents[0] != -1 or (parents[1:] >= np.arange(1, self.J)).any():
            raise ValueError("BodyLayer: parents[0] must be -1 and parents[j] < j")
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32))
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights"):          # SMPL-named buffers (state_dict surface)
            self.register_buffer(k, f(t[k]))
        self.register_buffer("parents", torch.as_tensor(parents.astype(np.int32)))
        # kernel-side layouts: joint regression folded into the tables, vertex-fastest blend shapes with a padded pitch
        self.VP = (self.NV + 63) // 64 * 64
        vp = lambda a: np.ascontiguousarray(np.pad(a, [(0, 0)] * (a.ndim - 1) + [(0, self.VP - self.NV)]), np.float32)
        jr = t["J_regressor"].astype(np.float64)
        self.register_buffer("_jt", f(jr @ t["v_template"].astype(np.float64)), persistent=False)                          # [J,3]
        self.register_buffer("_jsd", f(np.einsum("jv,vck->jck", jr, t["shapedirs"].astype(np.float64))), persistent=False)  # [J,3,nb]
        self.register_buffer("_vt", f(vp(t["v_template"].T)), persistent=False)                                             # [3][VP]
        self.register_buffer("_vsd", f(vp(t["shapedirs"].transpose(2, 1, 0))), persistent=False)                            # [nb][3][VP]
        self.register_buffer("_vpd", f(vp(t["posedirs"].transpose(2, 1, 0))), persistent=False)                             # [9(J-1)][3][VP]
        self.register_buffer("_vw", f(vp(t["weights"].T)), persistent=False)                                                # [J][VP]

    def _split_tables(self, dev):
        """the vertex tables as bf16 pieces in MFMA operand order (19 MB for SMPL), made on first use per device; the tables are fixed buffers"""
        key = (dev.type, dev.index, self._vt.data_ptr())
        if getattr(self, "_split_key", None) != key:
            L = _lib.lib()
            sp = torch.empty(L.mhe_lbs_split_floats(self.J, self.nb, self.VP), device=dev, dtype=torch.float32)
            ops.check(L.mhe_lbs_split_tables_f32(ops._ptr(self._vt), ops._ptr(self._vsd), ops._ptr(self._vpd), ops._ptr(self._vw), ops._ptr(sp),
                                                 self.J, self.nb, self.VP, ops._stream()), "mhe_lbs_split_tables_f32")
            self._split, self._split_key = sp, key
        return self._split

    def forward(self, betas, rotmats=None, pose6d=None, scale=1.0, want_verts=True):
        """betas (R,nb); rotmats (R,J,3,3) or pose6d (R,6J) -> {'vertices' (R,NV,3), 'joints' (R,J,3), 'rotmats'}"""
        if rotmats is None:
            rotmats = rot6d_to_rotmat(pose6d.reshape(-1, self.J, 6).contiguous())
        R = rotmats.shape[0]
        rotmats, betas = rotmats.contiguous(), betas.contiguous()
        ops._chk(rotmats, torch.float32, "body.rotmats", (R, self.J, 3, 3)); ops._chk(betas, torch.float32, "body.betas", (R, self.nb))
        L, dev = _lib.lib(), rotmats.device
        ws = torch.empty(L.mhe_lbs_workspace_floats(R, self.J, self.nb), device=dev, dtype=torch.float32)
        joints = torch.empty(R, self.J, 3, device=dev, dtype=torch.float32)
        ops.check(L.mhe_lbs_pose_f32(ops._ptr(rotmats), ops._ptr(betas), ops._ptr(self._jt), ops._ptr(self._jsd), ops._ptr(self.parents), ops._ptr(ws),
                                     ops._ptr(joints), R, self.J, self.nb, ops._stream()), "mhe_lbs_pose_f32")
        out = {"joints": joints, "rotmats": rotmats}
        if want_verts:
            verts = torch.empty(R, self.NV, 3, device=dev, dtype=torch.float32)
            if os.environ.get("MHE_LBS_MFMA", "1") == "1" and self.VP % 32 == 0 and L.mhe_lbs_skin_mfma_supported(R, self.J, self.nb, self.NV, self.VP):
                # both products on the matrix cores from bf16 pieces of the f32 operands (csrc/lbs_skin.hip); the table pieces are made once
                ops.check(L.mhe_lbs_skin_mfma_f32(ops._ptr(ws), ops._ptr(self._split_tables(dev)), ops._ptr(verts), R, self.J, self.nb, self.NV, self.VP,
                                                  float(scale), ops._stream()), "mhe_lbs_skin_mfma_f32")
                out["vertices"] = verts
                return out
            ops.check(L.mhe_lbs_skin_f32(ops._ptr(ws), ops._ptr(self._vt), ops._ptr(self._vsd), ops._ptr(self._vpd), ops._ptr(self._vw), ops._ptr(verts),
                                         R, self.J, self.nb, self.NV, self.VP, float(scale), ops._stream()), "mhe_lbs_skin_f32")
            out["vertices"] = verts
        return out


class BodyFlowHead(nn.Module):
    """ProHMR's sampling surface (reference README.md:26-42): `flow(conditioning_feats, num_samples)` draws K poses with
    log-probabilities from a ConditionalGlow over the 144-D 6D pose (features 144, hidden 1024, 4 layers x 2 blocks, context
    2048: SURVEY.md appendix A5), decoded by the body layer.  Parity unpinned (ProHMR's SMPLFlow is out of tree)."""
    def __init__(self, tables, context_features=2048, hidden=1024, num_layers=4, num_blocks=2):
        super().__init__()
        from .glow import ConditionalGlow
        self.body = BodyLayer(tables)
        self.flow = ConditionalGlow(6 * self.body.J, hidden, num_layers, num_blocks, context_features=context_features,
                                    dropout_probability=0.0)

    def forward(self, feats, num_samples, betas=None, noise=None, hyp_slice=None, want_verts=True):
        """feats (B,F) -> pose6d (B,K,6J), log_prob (B,K), vertices (B,K,NV,3), joints (B,K,J,3); hyp_slice = (lo, hi) decodes only
        hypotheses lo..hi-1 of every image (the hypothesis-sharded form)"""
        B = feats.shape[0]
        pose, logp, _ = self.flow.sample_and_log_prob(num_samples, noise=noise, context=feats)
        lo, hi = hyp_slice if hyp_slice is not None else (0, num_samples)
        p = pose[:, lo:hi].reshape(B * (hi - lo), -1).contiguous()
        bt = betas if betas is not None else torch.zeros(B, self.body.nb, device=feats.device)
        bt = bt[:, None, :].expand(B, hi - lo, self.body.nb).reshape(B * (hi - lo), self.body.nb).contiguous()
        out = self.body(bt, pose6d=p, want_verts=want_verts)
        res = {"pose6d": pose, "log_prob": logp, "joints": out["joints"].view(B, hi - lo, self.body.J, 3)}
        if want_verts:
            res["vertices"] = out["vertices"].view(B, hi - lo, self.body.NV, 3)
        return res
