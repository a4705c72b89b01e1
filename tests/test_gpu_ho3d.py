"""GPU: the HO3D input pipeline (csrc/ho3d.hip through mhentropy_amd/ho3d_dataloader.py) against the
fixtures the reference's own `Generate_ho3d_uv.__getitem__` produced (tests/golden/ho3d_*.npz) and against the oracle on a batch."""
import numpy as np
import pytest
import torch

from conftest import assert_close, check_ho3d_against_fixture, load_golden
from mhentropy_amd import synth

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else t


@pytest.mark.parametrize("case", range(6))
def test_pipeline_matches_reference_fixtures(gpu_lib, case):
    """pixels, masks and visibility bit-exact (integer work), floating-point targets to 2e-6 of their scale"""
    from mhentropy_amd import ho3d_dataloader as hd
    g = load_golden(f"ho3d_{case}")
    smp = synth.ho3d_sample(int(g["seed"]), tuple(float(v) for v in g["offset"]))
    aug = None
    if int(g["aug"]):
        p = synth.ho3d_aug_params(int(g["aug_seed"]))
        aug = np.array([[*p["pn"], p["scale"], p["angle"], p["tx"], p["ty"]]])
    img, t = hd.HO3DBatchPipeline()(hd.collate_decoded([smp]), aug)
    one = {k: _np(v)[0] for k, v in t.items() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 1}
    check_ho3d_against_fixture(_np(img)[0], one, g, f"ho3d case {case}", tol=2e-6)


def test_batch_against_the_oracle_and_feeds_the_model(gpu_lib):
    """a batch of 12 different samples (half of them augmented differently is not possible in one call: the flag is per batch, as in
    the reference's dataset object) against oracle/ho3d_ref.getitem sample by sample, evaluation and training mode; the output has the
    keys and shapes MHEnt.get_loss consumes (crop_uv, vis) and runs through it"""
    from mhentropy_amd import ho3d_dataloader as hd, harness
    from oracle import ho3d_ref
    offs = [(0, 0), (250, -180), (-280, 200), (120, 60), (-100, -150), (60, 190)]
    smps = [synth.ho3d_sample(10 + i, offs[i % 6], n_obj=1000 + 37 * i) for i in range(12)]
    pipe = hd.HO3DBatchPipeline()
    raw = hd.collate_decoded(smps)
    rs = np.random.RandomState(5)
    augs = hd.draw_aug(12, rs)
    for aug in (None, augs):
        img, t = pipe(raw, aug)
        assert img.shape == (12, 3, 256, 256) and t["crop_uv"].shape == (12, 42) and t["verts"].shape == (12, 2334)
        for i, s in enumerate(smps):
            prm = None if aug is None else {"pn": aug[i, :3], "scale": aug[i, 3], "angle": aug[i, 4], "tx": aug[i, 5], "ty": aug[i, 6]}
            oi, ot = ho3d_ref.getitem(s, prm)
            assert np.array_equal(_np(img[i]), oi), f"sample {i}: image"
            assert np.array_equal(_np(t["hand_mask"][i]), ot["hand_mask"]) and np.array_equal(_np(t["object_mask"][i]), ot["object_mask"])
            assert np.array_equal(_np(t["vis"][i]), ot["vis"]), f"sample {i}: vis"
            for k in ("crop_uv", "depth", "original_pose3d", "verts", "pose3d", "st", "scale", "crop_center", "crop_size", "pose3d_root",
                      "rot_mat_inv", "_rot_mat", "uvd"):
                assert_close(_np(t[k][i]).reshape(ot[k].shape), ot[k], 2e-6, 1e-7, what=f"sample {i}: {k}")
            n = s["obj_verts"].shape[0]
            assert_close(_np(t["object_verts"][i]).reshape(-1, 3)[:n], ot["object_verts_all"], 2e-6, 1e-4, what=f"sample {i}: object vertices")
    model = harness.build_mhent(backbone="resnet18", h_dims=(64, 64), num_steps=2, tables=synth.mano_tables(0)).cuda().train()
    out = model.get_loss(img, {k: t[k] for k in ("crop_uv", "vis")}, mods=["uv"], N=4)
    assert torch.isfinite(out["log_p"]).all() and out["log_p"].shape == (12,)


def test_properties_at_bench_batch(gpu_lib):
    """B = 256 (one C2 batch): identity augmentation parameters (unit colour factors, scale 1, angle 0, no shift) reproduce the
    evaluation-mode image exactly; pixels outside the rotated crop are the border value (-1 after normalisation); uv of visible
    joints lies within 4 pixels of the crop"""
    from mhentropy_amd import ho3d_dataloader as hd
    B = 256
    base = [synth.ho3d_sample(20 + i, ((i * 37) % 400 - 200, (i * 53) % 300 - 150)) for i in range(8)]
    raw = hd.collate_decoded([base[i % 8] for i in range(B)])
    pipe = hd.HO3DBatchPipeline()
    img0, t0 = pipe(raw)
    ident = np.tile(np.array([[1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0]]), (B, 1))
    img1, t1 = pipe(raw, ident)
    assert torch.equal(img0, img1) and torch.equal(t0["hand_mask"], t1["hand_mask"]) and torch.equal(t0["vis"], t1["vis"])
    assert_close(_np(t1["crop_uv"]), _np(t0["crop_uv"]), 1e-6, what="identity augmentation: uv")
    aug = hd.draw_aug(B, np.random.RandomState(3))
    img2, t2 = pipe(raw, aug)
    assert torch.isfinite(img2).all() and float(img2.min()) >= -1 and float(img2.max()) <= 1
    corner = img2[:, :, 0, 0]                 # scale <= 1 about the centre + rotation: a crop corner maps outside for most draws
    assert (corner == -1).all(1).float().mean() > 0.5
    uv_pix = (t2["crop_uv"].view(B, 21, 2) + 1) * 128
    v = t2["vis"] > 0
    assert bool(((uv_pix[v] >= -4.001) & (uv_pix[v] <= 259.001)).all())
