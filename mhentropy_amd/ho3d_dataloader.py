"""HO3D input pipeline on the GPU: the per-sample work of the reference's CPU workers
(hand/dataloader/ho3d_dataloader.py:272-459 `Generate_ho3d_uv.__getitem__`, `compute_st` hand/dataloader/rhddataloader.py:237-269)
and the default collate, from the DECODED arrays on, as two kernel launches per batch (csrc/ho3d.hip).  File reading and JPEG / PNG
decoding stay with the host workers; what they hand over is what imageio / cv2.imread / the annotation pickle return.

    pipe = HO3DBatchPipeline()
    image, target = pipe(collate_decoded(samples), aug=draw_aug(len(samples)))       # training: the reference's augmentation
    image, target = pipe(collate_decoded(samples))                                  # evaluation

The reference draws its augmentation parameters from np.random inside each worker; here they are an explicit [B, 7] float64
array (draw_aug uses the reference's distributions) so that a run is reproducible and checkable against the oracle.
There is no CPU path: the kernels are the implementation."""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops

HO3D2RHD = [0, 16, 15, 14, 13, 17, 3, 2, 1, 18, 6, 5, 4, 19, 12, 11, 10, 20, 9, 8, 7]        # hand/dataloader/ho3d_dataloader.py:17


def collate_decoded(samples, device="cuda"):
    """list of decoded samples {image u8 [480,640,3], depth_png u8 [480,640,3] (BGR, as cv2.imread returns the depth PNG), seg u8
    [120,160,3], joints3d [21,3] m, mesh [778,3] m, cam [3,3], obj_rot [3], obj_trans [3], obj_verts [n,3]} -> batched device
    tensors (object vertices padded to the longest, with counts)"""
    B = len(samples)
    nv = max(s["obj_verts"].shape[0] for s in samples)
    ov = np.zeros((B, nv, 3), np.float32)
    for i, s in enumerate(samples):
        ov[i, :s["obj_verts"].shape[0]] = s["obj_verts"]
    st = lambda k, dt: torch.as_tensor(np.stack([np.asarray(s[k], dt) for s in samples])).to(device)
    return {"image": st("image", np.uint8), "depth_png": st("depth_png", np.uint8), "seg": st("seg", np.uint8),
            "joints3d": st("joints3d", np.float32), "mesh": st("mesh", np.float32), "cam": st("cam", np.float32),
            "obj_rot": st("obj_rot", np.float32), "obj_trans": st("obj_trans", np.float32), "obj_verts": torch.as_tensor(ov).to(device),
            "obj_count": torch.as_tensor(np.asarray([s["obj_verts"].shape[0] for s in samples], np.int32)).to(device)}


def draw_aug(B, rng=None):
    """[B, 7] float64 = (colour factors x3, scale, angle, tx, ty) from the reference's distributions
    (hand/dataloader/ho3d_dataloader.py:162-176,191-194)"""
    rng = rng or np.random
    out = np.empty((B, 7))
    for b in range(B):          # the reference's order of draws per sample
        out[b, :3] = rng.uniform(0.6, 1.4, 3)
        out[b, 3] = rng.uniform(0.8, 1.0)
        out[b, 4] = 2 * np.pi * rng.rand(1)[0]
        out[b, 5] = np.maximum(np.minimum(rng.normal(0.0, 10.0), 40.0), -40.0)
        out[b, 6] = np.maximum(np.minimum(rng.normal(0.0, 10.0), 40.0), -40.0)
    return out


class HO3DBatchPipeline:
    def __init__(self, joint_idx="RHD", dpda="HO3D"):
        if joint_idx != "RHD" or dpda != "HO3D":
            raise NotImplementedError("the shipped configuration: RHD joint order, HO3D crop (hand/dataloader/ho3d_dataloader.py:201-216)")

    def __call__(self, raw, aug=None, object_idx=None):
        """raw: collate_decoded(...); aug: None (evaluation) or [B,7] float64 (numpy or device tensor); object_idx: optional [B,1000]
        int64 indices of the object vertices to keep (the reference draws them with np.random.choice).
        Returns (image [B,3,256,256] f32 in [-1,1], target dict with the reference's keys)."""
        L = _lib.lib()
        img = raw["image"]
        B, dev = img.shape[0], img.device
        ops._chk(img, torch.uint8, "ho3d.image", (B, 480, 640, 3)); ops._chk(raw["depth_png"], torch.uint8, "ho3d.depth_png", (B, 480, 640, 3))
        ops._chk(raw["seg"], torch.uint8, "ho3d.seg", (B, 120, 160, 3))
        ops._chk(raw["joints3d"], torch.float32, "ho3d.joints3d", (B, 21, 3)); ops._chk(raw["mesh"], torch.float32, "ho3d.mesh", (B, 778, 3))
        ops._chk(raw["cam"], torch.float32, "ho3d.cam", (B, 3, 3))
        for k in ("obj_rot", "obj_trans"):
            ops._chk(raw[k], torch.float32, "ho3d." + k, (B, 3))
        nv = raw["obj_verts"].shape[1]
        ops._chk(raw["obj_verts"], torch.float32, "ho3d.obj_verts", (B, nv, 3)); ops._chk(raw["obj_count"], torch.int32, "ho3d.obj_count", (B,))
        if aug is not None:
            aug = torch.as_tensor(np.asarray(aug, np.float64) if not torch.is_tensor(aug) else aug).to(dev).contiguous()
            ops._chk(aug, torch.float64, "ho3d.aug", (B, 7))
        f = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        t = {"crop_uv": f(B, 42), "vis": f(B, 21), "original_pose3d": f(B, 21, 3), "verts": f(B, 2334), "pose3d": f(B, 63), "st": f(B, 3),
             "scale": f(B), "crop_center": f(B, 2), "crop_size": f(B), "pose3d_root": f(B, 3), "rot_mat_inv": f(B, 3, 2), "_rot_mat": f(B, 2, 2),
             "uvd": f(B, 63)}
        obj_all = f(B, nv, 3)
        geom = torch.empty(B, L.mhe_ho3d_geom_doubles(), device=dev, dtype=torch.float64)
        p = ops._ptr
        ops.check(L.mhe_ho3d_targets(p(raw["joints3d"]), p(raw["mesh"]), p(raw["cam"]), p(raw["obj_rot"]), p(raw["obj_trans"]), p(raw["obj_verts"]),
                                     p(raw["obj_count"]), nv, p(raw["seg"]), p(raw["depth_png"]), p(aug), p(t["crop_uv"]), p(t["vis"]),
                                     p(t["original_pose3d"]), p(t["verts"]), p(t["pose3d"]), p(t["st"]), p(t["scale"]), p(t["crop_center"]),
                                     p(t["crop_size"]), p(t["pose3d_root"]), p(t["rot_mat_inv"]), p(t["_rot_mat"]), p(t["uvd"]), p(obj_all), p(geom),
                                     B, ops._stream()), "mhe_ho3d_targets")
        image = f(B, 3, 256, 256)
        hm = torch.empty(B, 256, 256, device=dev, dtype=torch.uint8)
        om = torch.empty_like(hm)
        depth = f(B, 256, 256)
        ops.check(L.mhe_ho3d_images(p(img), p(raw["seg"]), p(raw["depth_png"]), p(geom), p(aug), p(image), p(hm), p(om), p(depth), B, ops._stream()),
                  "mhe_ho3d_images")
        t.update(hand_mask=hm.bool(), object_mask=om.bool(), depth=depth, _root_idx=12, patch=torch.zeros(B, 3, device=dev),
                 hand_side=torch.zeros(B, device=dev), bone_length=t["scale"], camera=raw["cam"], dataset=["ho3d"] * B)
        t["object_verts"] = (obj_all if object_idx is None else
                             torch.gather(obj_all, 1, torch.as_tensor(object_idx, device=dev)[..., None].expand(-1, -1, 3))).flatten(-2)
        # hand/dataloader/dataset_transforms.py:13-14,35 (target_transform for 'ho3d')
        t["target_uvd_weight"] = torch.ones_like(t["pose3d"])
        t["image"] = image
        return image, t
