"""MANO decoder with the reference wrapper's API (reference hand/ManoLayer.py:10-165,
backed by hand/manopth/manolayer.py) on the fused HIP kernels (csrc/mano.hip).

Buffers are registered under `mano_layer.th_*` exactly as the reference's nested
manopth layer does, so checkpoints keep loading.  The MANO model file is
licence-restricted: pass `tables=` (a dict like mhentropy_amd.synth.mano_tables())
or point MANO_dir at a directory holding MANO_RIGHT.pkl.
"""
import os
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import ops, mano_pack

FreiHand2RHD_skeidx = [0, 4, 3, 2, 1, 8, 7, 6, 5, 12, 11, 10, 9, 16, 15, 14, 13, 20, 19, 18, 17]   # reference hand/utils.py:15


def _load_mano_pkl(path):
    with open(path, "rb") as f:
        d = pickle.load(f, encoding="latin1")
    g = lambda k: np.asarray(getattr(d[k], "r", d[k]))
    jr = d["J_regressor"]
    return {"shapedirs": g("shapedirs"), "posedirs": g("posedirs"), "v_template": g("v_template"),
            "J_regressor": np.asarray(jr.toarray() if hasattr(jr, "toarray") else jr), "weights": g("weights"),
            "hands_components": np.asarray(d["hands_components"]), "hands_mean": np.asarray(d["hands_mean"]),
            "betas": np.zeros(10, np.float32), "faces": np.asarray(d["f"]).astype(np.int64)}


class _ManoBuffers(nn.Module):
    """buffer holder named like manopth.ManoLayer (manolayer.py:69-101)"""
    def __init__(self, t, flat_hand_mean, ncomps):
        super().__init__()
        f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
        self.register_buffer("th_betas", f(t["betas"]).unsqueeze(0))
        self.register_buffer("th_shapedirs", f(t["shapedirs"]))
        self.register_buffer("th_posedirs", f(t["posedirs"]))
        self.register_buffer("th_v_template", f(t["v_template"]).unsqueeze(0))
        self.register_buffer("th_J_regressor", f(t["J_regressor"]))
        self.register_buffer("th_weights", f(t["weights"]))
        self.register_buffer("th_faces", torch.as_tensor(np.asarray(t["faces"]).astype(np.int32)).long())
        mean = np.zeros(45, np.float32) if flat_hand_mean else np.asarray(t["hands_mean"], np.float32)
        self.register_buffer("th_hands_mean", f(mean).unsqueeze(0))
        self.register_buffer("th_comps", f(t["hands_components"]))
        self.register_buffer("th_selected_comps", f(np.asarray(t["hands_components"])[:ncomps]))


class ManoLayer(nn.Module):
    def __init__(self, MANO_dir="./mano/", flat_hand_mean=True, ncomps=45, use_pca=False, n_latent=None,
                 skeidx="FreiHand", output_size=256, mask_sz=256, tables=None):
        super().__init__()
        if not use_pca or ncomps != 45:
            raise NotImplementedError("the hot path is built for use_pca=True, ncomps=45 "
                                      "(reference hand/CrossModalHand.py:72-74)")
        if n_latent is not None:
            raise NotImplementedError("latent->MANO regressors belong to the non-integrated baselines (out of scope)")
        if skeidx != "RHD":
            raise NotImplementedError("MHEnt builds the decoder with skeidx='RHD' (reference hand/network.py:360-363)")
        if tables is None:
            path = os.path.join(MANO_dir, "MANO_RIGHT.pkl")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found: the MANO model is licence-restricted; pass tables=")
            tables = _load_mano_pkl(path)
        self.mano_layer = _ManoBuffers(tables, flat_hand_mean, ncomps)
        self.skeidx, self.output_size, self.mask_sz, self.n_latent = skeidx, output_size, mask_sz, n_latent
        self._blob = None

    @property
    def Jreg(self):
        return self.mano_layer.th_J_regressor

    @property
    def mano_faces(self):
        return self.mano_layer.th_faces

    def table_blob(self):
        m = self.mano_layer
        key = (str(m.th_posedirs.device), m.th_posedirs._version, m.th_hands_mean._version)
        if self._blob is None or self._blob[0] != key:
            n = lambda t: t.detach().cpu().numpy()
            blob = mano_pack.pack_tables(n(m.th_shapedirs), n(m.th_posedirs), n(m.th_v_template)[0], n(m.th_J_regressor),
                                         n(m.th_weights), n(m.th_selected_comps), n(m.th_hands_mean)[0])
            self._blob = (key, torch.from_numpy(blob).to(m.th_posedirs.device))
        return self._blob[1]

    def forward(self, z=None, beta=None, theta=None):
        """reference hand/ManoLayer.py:45-60 -> {'beta','theta','mesh','joints','mano_joints'}"""
        if beta is None or theta is None:
            raise NotImplementedError("latent->MANO regressors are out of scope; pass beta= and theta=")
        beta, theta = beta.reshape(-1, 10).contiguous(), theta.reshape(-1, 48).contiguous()
        R = beta.shape[0]
        det = torch.zeros(R, 16, device=beta.device, dtype=torch.float32)
        det[:, :3], det[:, 3:13] = theta[:, :3], beta
        blob = self.table_blob()
        o = ops.mano_joints(theta[:, 3:].contiguous(), det, blob, want=("joints_mm", "mesh_mm"))
        mesh = o["mesh_mm"]
        return {"beta": beta, "theta": theta, "mesh": mesh, "joints": ops.mano_regress_joints(mesh, blob),
                "mano_joints": o["joints_mm"].view(R, 21, 3)}

    @staticmethod
    def batch_orth_proj(joint, scale_camera, trans_camera, image_size: int = 256, inv_norm=True):
        """reference hand/ManoLayer.py:150-165 (host-side helper for callers outside the fused path)."""
        out = scale_camera[:, None, :] * joint[:, :, :2] + trans_camera[:, None, :]
        if inv_norm:
            out = (out + 1.0) / 2.0 * image_size
        return out

    def render(self, *a, **k):
        raise NotImplementedError("the reference's renderer is commented out (hand/ManoLayer.py:40); never on the hot path")
