"""Deterministic synthetic data for the multi-hypothesis hot path.

The MANO model file (MANO_RIGHT.pkl) is licence-restricted and absent, the HO3D
dataset and the released checkpoint are not reachable, so
every test, fixture and benchmark runs on tensors generated here from a numpy
seed.  Shapes follow the buffers the reference registers
(reference hand/manopth/manolayer.py:69-108) and the target dict its data
loader emits (reference hand/dataloader/ho3d_dataloader.py:427-459).

This is data generation only - no arithmetic of the hot path lives here - so
both the product package and the oracle may use it.
"""
import numpy as np

N_VERTS = 778
N_FACES = 1538
N_JOINTS16 = 16
N_POSE_BASIS = 135
N_SHAPE = 10
N_PCA = 45

# MANO kinematic tree: parent of each of the 16 joints (root has none).  The
# reference's level-wise chain (manolayer.py:197-199) encodes the same tree.
MANO_PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)


def mano_tables(seed=0):
    """Synthetic MANO-shaped tables as float32/int64 numpy arrays.

    Keys and shapes mirror the buffers of the reference layer
    (manolayer.py:69-108): shapedirs (778,3,10), posedirs (778,3,135),
    v_template (778,3), J_regressor (16,778), weights (778,16),
    hands_components (45,45), hands_mean (45,), faces (1538,3) int64.
    """
    rng = np.random.default_rng(seed)
    t = {}
    t["shapedirs"] = rng.normal(0.0, 1e-3, (N_VERTS, 3, N_SHAPE)).astype(np.float32)
    t["posedirs"] = rng.normal(0.0, 1e-3, (N_VERTS, 3, N_POSE_BASIS)).astype(np.float32)
    t["v_template"] = rng.normal(0.0, 0.05, (N_VERTS, 3)).astype(np.float32)
    # sparse-ish positive joint regressor, rows sum to one
    jr = rng.random((N_JOINTS16, N_VERTS)) ** 8
    jr[jr < 0.2] = 0.0
    jr[np.arange(N_JOINTS16), rng.integers(0, N_VERTS, N_JOINTS16)] += 0.5
    jr /= jr.sum(1, keepdims=True)
    t["J_regressor"] = jr.astype(np.float32)
    # skinning weights: positive, rows sum to one, a few dominant joints
    w = rng.random((N_VERTS, N_JOINTS16)) ** 6
    w /= w.sum(1, keepdims=True)
    t["weights"] = w.astype(np.float32)
    q, _ = np.linalg.qr(rng.normal(size=(N_PCA, N_PCA)))
    t["hands_components"] = (q * rng.uniform(0.2, 1.0, (N_PCA, 1))).astype(np.float32)
    t["hands_mean"] = rng.normal(0.0, 0.3, (N_PCA,)).astype(np.float32)
    t["betas"] = np.zeros((N_SHAPE,), np.float32)
    t["faces"] = rng.integers(0, N_VERTS, (N_FACES, 3)).astype(np.int64)
    kt = np.zeros((2, N_JOINTS16), np.int64)
    kt[0] = np.array(MANO_PARENTS, np.int64)
    kt[0, 0] = 4294967295  # the real file stores uint32(-1) for the root
    kt[1] = np.arange(N_JOINTS16)
    t["kintree_table"] = kt
    return t


def _linear(rng, out_f, in_f):
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) - the distribution torch.nn.Linear
    uses by default (the reference never calls RealNVP._init, flows.py:200)."""
    b = 1.0 / np.sqrt(in_f)
    return (rng.uniform(-b, b, (out_f, in_f)).astype(np.float32),
            rng.uniform(-b, b, (out_f,)).astype(np.float32))


def flow_state(seed=0, dim=45, cond_dim=512, h_dims=(512, 512), num_steps=6):
    """state_dict (numpy) of a conditional RealNVP with the reference's key
    names (flows.py:86-95,188-195): mask, {s,t}.{i}.l.{0,1,2}.{weight,bias},
    {s,t}.{i}.c.{0,1}.{weight,bias}."""
    rng = np.random.default_rng(seed)
    a = np.array([0] * (dim // 2) + [1] * (dim - dim // 2), np.float32)
    mask = np.stack([a, 1 - a] * num_steps).astype(np.float32)
    sd = {"mask": mask}
    dims = [(h_dims[0], dim), (h_dims[1], h_dims[0]), (dim, h_dims[1])]
    for net in ("t", "s"):
        for i in range(2 * num_steps):
            for j, (o, n) in enumerate(dims):
                w, b = _linear(rng, o, n)
                sd[f"{net}.{i}.l.{j}.weight"], sd[f"{net}.{i}.l.{j}.bias"] = w, b
            for j, h in enumerate(h_dims):
                w, b = _linear(rng, h, cond_dim)
                sd[f"{net}.{i}.c.{j}.weight"], sd[f"{net}.{i}.c.{j}.bias"] = w, b
    return sd


def head_state(seed=0, feat_dim=2048, n_latent=512, det_out=16):
    """l1/l2 heads of BasicEnc (network.py:87-88) and det_head (network.py:380-383)."""
    rng = np.random.default_rng(seed + 1000)
    sd = {}
    sd["feat_extractor.l1.0.weight"], sd["feat_extractor.l1.0.bias"] = _linear(rng, n_latent, feat_dim)
    sd["feat_extractor.l2.0.weight"], sd["feat_extractor.l2.0.bias"] = _linear(rng, n_latent, feat_dim)
    sd["det_head.0.weight"], sd["det_head.0.bias"] = _linear(rng, n_latent, n_latent)
    w, b = _linear(rng, det_out, n_latent)
    # keep the deterministic head's outputs in the range the priors expect:
    # beta within +-0.03 (network.py:435), log-scale and translation small
    sd["det_head.2.weight"], sd["det_head.2.bias"] = w * 0.05, b * 0.05
    return sd


RESNET_CFG = {
    "resnet18": ("basic", (2, 2, 2, 2), 512),
    "resnet50": ("bottleneck", (3, 4, 6, 3), 2048),
}


def resnet_state(seed=0, arch="resnet50"):
    """Random-init ResNet v1.5 state_dict with torchvision's key names
    (reference hand/network.py:54-61 builds torchvision.models.resnet18/50 and
    sets fc = Identity).  He-normal conv weights, BN gamma ~ U(0.5,1.5)."""
    rng = np.random.default_rng(seed + 2000)
    kind, blocks, _ = RESNET_CFG[arch]
    sd = {}

    def conv(name, o, i, k):
        std = np.sqrt(2.0 / (i * k * k))
        sd[name + ".weight"] = rng.normal(0, std, (o, i, k, k)).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = rng.uniform(0.5, 1.5, (c,)).astype(np.float32)
        sd[name + ".bias"] = rng.normal(0, 0.1, (c,)).astype(np.float32)
        sd[name + ".running_mean"] = rng.normal(0, 0.1, (c,)).astype(np.float32)
        sd[name + ".running_var"] = rng.uniform(0.5, 1.5, (c,)).astype(np.float32)
        sd[name + ".num_batches_tracked"] = np.zeros((), np.int64)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    inplanes = 64
    exp = 4 if kind == "bottleneck" else 1
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), blocks)):
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 0) else 1
            p = f"layer{li + 1}.{bi}"
            if kind == "bottleneck":
                conv(p + ".conv1", planes, inplanes, 1); bn(p + ".bn1", planes)
                conv(p + ".conv2", planes, planes, 3); bn(p + ".bn2", planes)
                conv(p + ".conv3", planes * 4, planes, 1); bn(p + ".bn3", planes * 4)
            else:
                conv(p + ".conv1", planes, inplanes, 3); bn(p + ".bn1", planes)
                conv(p + ".conv2", planes, planes, 3); bn(p + ".bn2", planes)
            if stride != 1 or inplanes != planes * exp:
                conv(p + ".downsample.0", planes * exp, inplanes, 1)
                bn(p + ".downsample.1", planes * exp)
            inplanes = planes * exp
    return sd


def batch(seed=0, B=2, image_size=256, with_image=True):
    """Synthetic (image, target) pair with the keys the hot path consumes
    (SURVEY.md section 8a row a0 / 8d): image (B,3,S,S) in [-1,1], crop_uv (B,42)
    in [-1,1], vis (B,21) in {0,1}, st (B,3), pose3d (B,63), scale (B,)."""
    rng = np.random.default_rng(seed + 3000)
    y = {}
    if with_image:
        x = np.clip(rng.normal(0, 0.5, (B, 3, image_size, image_size)), -1, 1).astype(np.float32)
    else:
        x = None
    y["crop_uv"] = rng.uniform(-1, 1, (B, 42)).astype(np.float32)
    y["vis"] = (rng.random((B, 21)) < 0.7).astype(np.float32)
    st = np.concatenate([rng.uniform(0.5, 1.5, (B, 1)), rng.uniform(-0.2, 0.2, (B, 2))], 1)
    y["st"] = st.astype(np.float32)
    y["pose3d"] = rng.normal(0, 1, (B, 63)).astype(np.float32)
    y["scale"] = rng.uniform(0.5, 1.5, (B,)).astype(np.float32)
    return x, y


def structured_images(seed=0, B=2, image_size=256):
    """Images with image-to-image structure (a smooth random field per image: random 4x4 / 16x16 colour patterns bilinearly enlarged, a
    per-image gain and offset, a little pixel noise), in [-1,1].  batch()'s default images are i.i.d. white noise: after global average pooling
    every image of a batch has almost the same feature, so the part of the gradient that survives train-mode BatchNorm is a small residual
    of a large batch-constant term - a badly conditioned problem for ANY reduced-precision forward (tools/bf16_grad_sensitivity.py).  The
    gradient-parity tests of the bf16 mode use these images for a well-conditioned case next to the bench's own workload."""
    rng = np.random.default_rng(seed + 3500)

    def up(a, S):                                   # bilinear enlargement of (B,3,h,h) to (B,3,S,S), align_corners=False
        h = a.shape[-1]
        c = (np.arange(S) + 0.5) * h / S - 0.5
        i0 = np.clip(np.floor(c).astype(np.int64), 0, h - 1)
        i1 = np.clip(i0 + 1, 0, h - 1)
        w = np.clip(c - i0, 0.0, 1.0).astype(np.float32)
        rows = a[:, :, i0, :] * (1 - w)[None, None, :, None] + a[:, :, i1, :] * w[None, None, :, None]
        return rows[:, :, :, i0] * (1 - w) + rows[:, :, :, i1] * w
    S = image_size
    x = 0.9 * up(rng.normal(0, 1, (B, 3, 4, 4)).astype(np.float32), S) + 0.5 * up(rng.normal(0, 1, (B, 3, 16, 16)).astype(np.float32), S)
    x = x * rng.uniform(0.3, 1.0, (B, 1, 1, 1)).astype(np.float32) + rng.normal(0, 0.3, (B, 3, 1, 1)).astype(np.float32)
    x = x + rng.normal(0, 0.05, x.shape).astype(np.float32)
    return np.clip(x, -1, 1).astype(np.float32)


def noise(seed, rows, dim=45):
    """Host-generated base noise z0 ~ N(0, I) (parity runs
    must feed captured noise, GPU and CPU generators differ)."""
    rng = np.random.default_rng(seed + 4000)
    return rng.normal(0, 1, (rows, dim)).astype(np.float32)


def glow_state(seed=0, features=45, hidden=512, num_layers=4, num_blocks=2, context_features=512):
    """state_dict (nflows key names) of a ConditionalGlow with well-conditioned random parameters: LU factors near
    identity, small last layers (nflows initialises them ~0) so that the inverse pass stays O(1)."""
    rng = np.random.default_rng(seed + 4000)
    f32 = lambda a: np.asarray(a, np.float32)
    sd = {}
    mask = np.ones(features); mask[::2] = -1
    for l in range(num_layers):
        p = f"_transform._transforms.{3 * l}."
        sd[p + "log_scale"], sd[p + "shift"] = f32(rng.normal(0, 0.1, features)), f32(rng.normal(0, 0.1, features))
        p = f"_transform._transforms.{3 * l + 1}."
        n = features * (features - 1) // 2
        sd[p + "lower_entries"], sd[p + "upper_entries"] = f32(rng.normal(0, 0.03, n)), f32(rng.normal(0, 0.03, n))
        sd[p + "unconstrained_upper_diag"], sd[p + "bias"] = f32(0.5 + rng.normal(0, 0.1, features)), f32(rng.normal(0, 0.1, features))
        p = f"_transform._transforms.{3 * l + 2}.transform_net."
        nid, nt = int((mask <= 0).sum()), int((mask > 0).sum())
        sd[p + "initial_layer.weight"], sd[p + "initial_layer.bias"] = _linear(rng, hidden, nid + context_features)
        for b in range(num_blocks):
            q = p + f"blocks.{b}."
            sd[q + "linear_layers.0.weight"], sd[q + "linear_layers.0.bias"] = _linear(rng, hidden, hidden)
            w, bb = _linear(rng, hidden, hidden)
            sd[q + "linear_layers.1.weight"], sd[q + "linear_layers.1.bias"] = w * 0.3, bb * 0.3
            sd[q + "context_layer.weight"], sd[q + "context_layer.bias"] = _linear(rng, hidden, context_features)
        w, bb = _linear(rng, 2 * nt, hidden)
        sd[p + "final_layer.weight"], sd[p + "final_layer.bias"] = w * 0.2, bb * 0.2
        mask = -mask
    return sd


def ho3d_sample(seed=0, offset=(0.0, 0.0), n_obj=1200):
    """A synthetic DECODED HO3D sample (what imageio / cv2 / pickle hand to hand/dataloader/ho3d_dataloader.py:279-291): random RGB
    frame, 3-channel depth PNG as cv2.imread returns it (BGR: value = R + 256 G), a 120x160 segmentation whose hand / object
    channels cover the projected hand / object, 21 hand joints and a 778-vertex cloud in OpenGL camera coordinates (metres, z < 0),
    intrinsics, an object pose and its vertices.  offset moves the hand in the image (towards a border: padded crops)."""
    rng = np.random.default_rng(seed + 7000)
    f32 = lambda a: np.asarray(a, np.float32)
    cam = f32([[617.3, 0, 312.4], [0, 617.1, 241.4], [0, 0, 1]])
    z = -rng.uniform(0.45, 0.65)
    centre = np.array([offset[0] * -z / 617.0, -offset[1] * -z / 617.0, z])
    joints = f32(centre + rng.normal(0, 0.035, (21, 3)) * [1, 1, 0.4])
    mesh = f32(joints[rng.integers(0, 21, 778)] + rng.normal(0, 0.008, (778, 3)))
    obj_verts = f32(rng.uniform(-0.06, 0.06, (n_obj, 3)))
    obj_rot = f32(rng.normal(0, 0.8, 3))
    obj_trans = f32(centre + rng.normal(0, 0.03, 3) * [1, 1, 0.3])
    image = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    # projected hand / object blobs (same projection as the reference: flip y, z then pinhole)
    def proj(p):
        q = p * [1, -1, -1]
        return np.stack([q[:, 0] * 617.3 / q[:, 2] + 312.4, q[:, 1] * 617.1 / q[:, 2] + 241.4], 1), q[:, 2]
    seg = np.zeros((120, 160, 3), np.uint8)
    depth_m = np.full((480, 640), 1.2)
    uv_h, z_h = proj(mesh)
    for (u, v), zz in zip(uv_h[::3], z_h[::3]):
        if rng.random() < 0.8:
            a, b = int(v) // 4, int(u) // 4
            seg[max(a - 1, 0):a + 2, max(b - 1, 0):b + 2, 2] = 255
            depth_m[max(int(v) - 6, 0):int(v) + 7, max(int(u) - 6, 0):int(u) + 7] = zz + rng.choice([-0.004, 0.0, 0.06])
    R = _rodrigues(obj_rot)
    uv_o, _ = proj((obj_verts @ R.T + obj_trans)[::7])
    for u, v in uv_o:
        a, b = int(v) // 4, int(u) // 4
        if 0 <= a < 120 and 0 <= b < 160:
            seg[a, b, 1] = 230
    d16 = np.clip(np.round(depth_m / 0.00012498664727900177), 0, 65535).astype(np.uint16)
    depth_png = np.zeros((480, 640, 3), np.uint8)
    depth_png[:, :, 2], depth_png[:, :, 1] = d16 & 255, d16 >> 8
    return {"image": image, "depth_png": depth_png, "seg": seg, "joints3d": joints, "mesh": mesh, "cam": cam, "obj_rot": obj_rot,
            "obj_trans": obj_trans, "obj_verts": obj_verts}


def _rodrigues(r):
    th = float(np.linalg.norm(r))
    if th < 1e-12:
        return np.eye(3)
    k = np.asarray(r, np.float64) / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * K


def ho3d_aug_params(seed):
    """the random draws of the reference's augmentation in its own order and distributions (ho3d_dataloader.py:162-176,191-194):
    three colour factors U(0.6, 1.4), scale U(0.8, 1), angle 2 pi U(0, 1), translations N(0, 10) clipped to +-40"""
    rs = np.random.RandomState(seed)
    pn = rs.uniform(0.6, 1.4, 3)
    scale = rs.uniform(low=0.8, high=1.0)
    angle = 2 * np.pi * rs.rand(1)[0]
    tx = np.maximum(np.minimum(rs.normal(0.0, 10.0), 40.0), -40.0)
    ty = np.maximum(np.minimum(rs.normal(0.0, 10.0), 40.0), -40.0)
    return {"pn": pn, "scale": float(scale), "angle": float(angle), "tx": float(tx), "ty": float(ty)}
