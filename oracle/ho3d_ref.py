"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement (numpy, float64 where numpy 2 promotes to it) of the reference's HO3D sample pipeline from the DECODED arrays on
(/root/reference/hand/dataloader/ho3d_dataloader.py:272-459 `Generate_ho3d_uv.__getitem__`; helpers :32-199; `compute_st`
/root/reference/hand/dataloader/rhddataloader.py:237-269 with `align_w_scale` /root/reference/hand/utils.py:502-525), SURVEY.md
section 8 row f4.  File reading / PNG-JPEG decoding stays on the host and is not part of it.

Pinning: oracle/gen_golden.py:gen_ho3d runs the reference's own `__getitem__` (module imported with the placeholder modules of
SURVEY.md A6) on synthetic samples, with its file readers pointed at the synthetic arrays and the OpenCV / torchvision calls it makes
served by the restatements below -> the reference's in-tree logic (projection, boxes, crop arithmetic, visibility loops,
augmentation bookkeeping, target assembly, compute_st) is PINNED by tests/golden/ho3d_*.npz; the third-party primitives are
restated from their published algorithms and are UNPINNED (OpenCV 4.x is not installed here):
  cv2.resize(INTER_NEAREST)            imgproc/resize.cpp: sx = min(floor(dx * src/dst), src - 1)
  cv2.copyMakeBorder(BORDER_CONSTANT)
  cv2.getRotationMatrix2D              imgproc/imgwarp.cpp: alpha = s cos a, beta = s sin a (a in degrees)
  cv2.warpAffine(INTER_NEAREST, BORDER_CONSTANT 0)  imgwarp.cpp: inverse map in 10-bit fixed point, round_delta = 512
  cv2.Rodrigues (vector -> matrix)     calib3d
  torchvision ToPILImage / ToTensor / Normalize(0.5, 0.5)
Random draws (np.random in the reference) enter as explicit parameters."""
import math

import numpy as np
from scipy.linalg import orthogonal_procrustes

HO3D2RHD = [0, 16, 15, 14, 13, 17, 3, 2, 1, 18, 6, 5, 4, 19, 12, 11, 10, 20, 9, 8, 7]          # ho3d_dataloader.py:17
DEPTH_SCALE = 0.00012498664727900177                                                        # ho3d_vis_utils.py:463
_FLIP = np.array([[1., 0., 0.], [0, -1., 0.], [0., 0., -1.]], dtype=np.float32)                 # ho3d_dataloader.py:33


# ---- third-party primitives (unpinned restatements) ---------------------------------------------------------------------
def resize_nearest(img, dsize):
    """cv2.resize(img, (W, H), interpolation=INTER_NEAREST)"""
    W, H = dsize
    h, w = img.shape[:2]
    sx = np.minimum(np.floor(np.arange(W) * (w / W)).astype(np.int64), w - 1)
    sy = np.minimum(np.floor(np.arange(H) * (h / H)).astype(np.int64), h - 1)
    return img[sy][:, sx]


def copy_make_border(img, top, bottom, left, right, value):
    pad = [(top, bottom), (left, right)] + [(0, 0)] * (img.ndim - 2)
    if img.ndim == 2:
        return np.pad(img, pad, constant_values=value[0])
    out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), img.dtype)
    out[...] = np.asarray(value[:img.shape[2]], img.dtype)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out


def get_rotation_matrix_2d(center, angle_deg, scale):
    a = angle_deg * math.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = center
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)


def invert_affine(M):
    """the inverse map warpAffine builds (doubles, same operation order)"""
    m = np.array(M, np.float64).reshape(-1).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def warp_source_index(M, W=256, H=256):
    """integer source coordinates (X, Y) [H, W] of cv2.warpAffine(.., M, (W, H), flags=INTER_NEAREST): 10-bit fixed point"""
    m = invert_affine(M)
    rnd = lambda v: np.rint(v).astype(np.int64)                       # saturate_cast<int>(double) = round half to even
    x = np.arange(W)
    adelta, bdelta = rnd(m[0] * x * 1024), rnd(m[3] * x * 1024)
    y = np.arange(H)
    X0 = rnd((m[1] * y + m[2]) * 1024) + 512
    Y0 = rnd((m[4] * y + m[5]) * 1024) + 512
    return (X0[:, None] + adelta[None, :]) >> 10, (Y0[:, None] + bdelta[None, :]) >> 10


def warp_affine_nearest(img, M, dsize=(256, 256)):
    X, Y = warp_source_index(M, *dsize)
    ok = (X >= 0) & (X < img.shape[1]) & (Y >= 0) & (Y < img.shape[0])
    out = np.zeros((dsize[1], dsize[0]) + img.shape[2:], img.dtype)
    out[ok] = img[Y[ok], X[ok]]
    return out


def rodrigues(r):
    r = np.asarray(r, np.float64).reshape(3)
    th = math.sqrt(float(r @ r))
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    c, s = math.cos(th), math.sin(th)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * Kx


def to_tensor_normalized(img_u8):
    """ToPILImage -> ToTensor -> Normalize([.5]*3, [.5]*3): HWC uint8 -> CHW float32"""
    t = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    return (t - np.float32(0.5)) / np.float32(0.5)


# ---- the reference's own helpers --------------------------------------------------------------------------------------------
def coord_change(xyz):
    return xyz.dot(_FLIP.T)                                             # ho3d_dataloader.py:32-35


def xyz2uvd(xyz, K):
    """ho3d_dataloader.py:72-80 (result stored as float32)"""
    xyz = xyz.dot(_FLIP.T)
    fx, fy, fu, fv = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    uvd = np.zeros_like(xyz, np.float32)
    uvd[:, 0] = xyz[:, 0] * fx / xyz[:, 2] + fu
    uvd[:, 1] = xyz[:, 1] * fy / xyz[:, 2] + fv
    uvd[:, 2] = xyz[:, 2]
    return uvd


def get_bbox_joints(j2d, factor):
    """ho3d_dataloader.py:82-92: integer-truncated centre, half extents scaled by factor"""
    mn, mx = j2d.min(0), j2d.max(0)
    c = np.asarray([int((mx[0] + mn[0]) / 2), int((mx[1] + mn[1]) / 2)])
    d = np.asarray([(mx[0] - mn[0]) * factor / 2, (mx[1] - mn[1]) * factor / 2])
    return np.array([*(c - d), *(c + d)], dtype=np.float32)


def fuse_bbox(b1, b2, img_shape):
    """ho3d_dataloader.py:94-108 (x is clamped with img_shape[0], y with img_shape[1], as written there)"""
    bb = np.concatenate((b1.reshape(2, 2), b2.reshape(2, 2)), 0)
    mn_x, mn_y = bb.min(0)
    mn_x, mn_y = max(0, mn_x), max(0, mn_y)
    mx_x, mx_y = bb.max(0)
    mx_x, mx_y = min(mx_x, img_shape[0]), min(mx_y, img_shape[1])
    c = np.asarray([int((mx_x + mn_x) / 2), int((mx_y + mn_y) / 2)])
    return c, max(mx_x - mn_x, mx_y - mn_y) * 1.0


def crop_window(center, size):
    """ho3d_dataloader.py:110-114: x1, y1, x2, y2"""
    return (int(np.round(center[0] - size)), int(np.round(center[1] - size)), int(np.round(center[0] + size)), int(np.round(center[1] + size)))


def imcrop(img, center, size):
    """ho3d_dataloader.py:110-138: crop, padding with 127 (3-channel) / 0 (single channel) where the window leaves the image"""
    x1, y1, x2, y2 = crop_window(center, size)
    if x1 < 0 or y1 < 0 or x2 > img.shape[1] or y2 > img.shape[0]:
        img = copy_make_border(img, -min(0, y1), max(y2 - img.shape[0], 0), -min(0, x1), max(x2 - img.shape[1], 0),
                               [0] if img.ndim < 3 else [127, 127, 127])
        y2 += -min(0, y1); y1 += -min(0, y1); x2 += -min(0, x1); x1 += -min(0, x1)
    return img[y1:y2, x1:x2]


def processing_pose3d(p, root=4, rel=5):
    """ho3d_dataloader.py:154-160"""
    r = p[root]
    prel = p - r
    bone = np.sqrt(np.sum(np.square(prel[root] - prel[rel])))
    return prel, prel / bone, r, bone


def compute_st(pose3d, crop_uv):
    """rhddataloader.py:237-269 + utils.py:502-525: uv ~ s * normed_xy + t from a scaled Procrustes fit (rotation ignored in t)"""
    m1 = np.array(crop_uv.reshape(-1, 2), np.float64)
    m2 = np.array(pose3d.reshape(-1, 3)[:, :2], np.float64)
    t1, t2 = m1.mean(0), m2.mean(0)
    a, b = m1 - t1, m2 - t2
    s1 = np.linalg.norm(a) + 1e-8
    s2 = np.linalg.norm(b) + 1e-8
    _, s = orthogonal_procrustes(a / s1, b / s2)
    t = -t2 / s2 * s * s1 + t1
    s *= s1 / s2
    return np.concatenate([np.array([s]), t])


# ---- one sample -----------------------------------------------------------------------------------------------------------------
def decode_depth(depth_png_bgr):
    """ho3d_vis_utils.py:457-469 with the arithmetic the reference's numpy (< 2) performed: 16-bit value * scale, float64"""
    return (depth_png_bgr[:, :, 2].astype(np.uint16) + depth_png_bgr[:, :, 1].astype(np.uint16) * 256) * DEPTH_SCALE


def getitem(s, aug=None, joint_idx="RHD"):
    """s: decoded sample {image u8 [480,640,3], depth_png u8 [480,640,3] (as cv2.imread returns it), seg u8 [120,160,3],
    joints3d [21,3] m, mesh [778,3] m, cam [3,3], obj_rot [3], obj_trans [3], obj_verts [n,3]};
    aug: None (evaluation) or {pn [3], scale, angle, tx, ty} = the reference's np.random draws (ho3d_dataloader.py:162-198).
    Returns (image [3,256,256] f32, target dict) as ho3d_dataloader.py:272-459 (dpda='HO3D')."""
    image, seg = s["image"], resize_nearest(s["seg"], (640, 480))
    depth = decode_depth(s["depth_png"])
    J = s["joints3d"] * 1000.0
    mesh = s["mesh"] * 1000.0
    R = rodrigues(s["obj_rot"])
    obj = (np.matmul(s["obj_verts"], R.T) + s["obj_trans"]) * 1000.0
    J_uvd, obj_uvd = xyz2uvd(J, s["cam"]), xyz2uvd(obj, s["cam"])
    J, mesh, obj = coord_change(J), coord_change(mesh), coord_change(obj)
    center, scale = fuse_bbox(get_bbox_joints(J_uvd[:, :2], 1.5), get_bbox_joints(obj_uvd[:, :2], 1.0), image.shape)
    size = scale / 2
    crop = lambda a: resize_nearest(imcrop(a, center, size), (256, 256))
    image_crop, depth_crop, seg_crop = crop(image), crop(depth), crop(seg)
    obj_mask, hand_mask_crop = seg_crop[:, :, 1] > 200, seg_crop[:, :, 2] > 200
    hand_mask = seg[:, :, 2] > 200
    uv = J_uvd[:, :2].copy()
    uv[:, 0] = (uv[:, 0] - center[0] + size) * (256.0 / (size * 2))
    uv[:, 1] = (uv[:, 1] - center[1] + size) * (256.0 / (size * 2))
    vis = np.zeros(21, bool)
    for i in range(21):                                                 # :367-384: any hand-mask pixel of the 9x9 window within 40 mm in front
        u0, v0, d = int(J_uvd[i, 0]), int(J_uvd[i, 1]), J_uvd[i, 2]
        for u in range(u0 - 4, u0 + 5):
            for v in range(v0 - 4, v0 + 5):
                if 0 <= u < 640 and 0 <= v < 480 and hand_mask[v, u] and (d - depth[v, u] * 1000) < 40:
                    vis[i] = True
    _, normed, _, bone = processing_pose3d(J)
    rot = np.eye(2, 3)
    if aug is not None:
        image_crop = image_crop.copy()
        for c in range(3):                                              # rgb_processing :191-198, stored back into the uint8 image
            image_crop[:, :, c] = np.minimum(255.0, np.maximum(0.0, image_crop[:, :, c] * aug["pn"][c]))
        rot = get_rotation_matrix_2d((128, 128), -180.0 * aug["angle"] / math.pi, aug["scale"])
        rot[0, 2] += aug["tx"]; rot[1, 2] += aug["ty"]
        ca, sa = math.cos(aug["angle"]), math.sin(aug["angle"])
        normed = normed.copy()
        normed[:, 0], normed[:, 1] = ca * normed[:, 0] - sa * normed[:, 1], sa * normed[:, 0] + ca * normed[:, 1]
        uv = np.dot(rot, np.concatenate([uv, np.ones((21, 1))], 1).T).T
        image_crop = warp_affine_nearest(image_crop, rot)
        hand_mask_crop = warp_affine_nearest(hand_mask_crop.astype(np.float32), rot).astype(bool)
        obj_mask = warp_affine_nearest(obj_mask.astype(np.float32), rot).astype(bool)
        depth_crop = warp_affine_nearest(depth_crop, rot)
    for i in range(21):                                                 # :396-409: the 9x9 window has a pixel inside the crop
        u, v = uv[i]
        if not any(0 <= u + du <= 255 and 0 <= v + dv <= 255 for du in range(-4, 5) for dv in range(-4, 5)):
            vis[i] = False
    img = to_tensor_normalized(image_crop.astype(np.uint8))
    if joint_idx == "RHD":
        uv, J, normed, vis = uv[HO3D2RHD], J[HO3D2RHD], normed[HO3D2RHD], vis[HO3D2RHD]
    uv = uv / 256 * 2 - 1
    rmi = np.eye(3)
    rmi[:2, :] = rot
    rmi = np.linalg.inv(rmi.T)[:, :2]
    f32 = lambda a: np.asarray(a, np.float32)
    t = {"crop_uv": f32(uv).reshape(-1), "hand_mask": hand_mask_crop, "object_mask": obj_mask, "vis": f32(vis), "depth": f32(depth_crop),
         "original_pose3d": f32(J), "verts": f32(mesh).reshape(-1), "pose3d": f32(normed).reshape(-1), "st": f32(compute_st(normed, uv)),
         "scale": f32(bone / 1000.), "crop_center": f32(center), "crop_size": f32(size), "pose3d_root": f32(J[12] / 1000),
         "camera": f32(s["cam"]), "rot_mat_inv": f32(rmi), "_rot_mat": f32(rot[:, :2] / np.linalg.norm(rot[0, :2])),
         "uvd": f32(np.concatenate([uv, normed[:, [-1]]], 1)).ravel(), "object_verts_all": f32(obj)}
    return img, t
