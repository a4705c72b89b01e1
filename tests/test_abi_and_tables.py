"""CPU: the C-ABI library loads and exports every symbol include/mhe.h declares;
host-side packing/layout constants agree with the kernels; the index tables baked
into the kernels equal the reference's gathers (bit-exact integer work)."""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from mhentropy_amd import _lib, mano_pack, synth
from oracle import mano_ref


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "mhe.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mhe_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mhe.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)
    assert L.mhe_abi_version() == 4


def test_layout_constants_agree():
    L = _lib.lib()
    assert L.mhe_mano_table_floats() == mano_pack.TOTAL_FLOATS
    hdr = open(os.path.join(ROOT, "mhentropy_amd", "csrc", "mano_layout.h")).read()
    for name in ("COMPS", "MEAN", "JT", "JSD", "TIP_T", "TIP_SD", "TIP_PD", "TIP_W", "JOINT_FLOATS", "VP", "NV"):
        v = int(re.search(rf"constexpr int {name} = (\d+);", hdr).group(1))
        assert v == getattr(mano_pack, name), name
    # stage geometry of the packed flow stream: 78 stages of 16 KiB for the shipped width
    assert L.mhe_flow_packed_floats_per_net(45, 512) == 78 * 4096
    assert L.mhe_flow_packed_floats_per_net(45, 64) == 3 * 4096
    assert L.mhe_flow_packed_floats_per_net(45, 100) == 0 and L.mhe_flow_packed_floats_per_net(49, 64) == 0


def _c_array(src, name):
    m = re.search(rf"{name}\[\d+\] = \{{([^}}]*)\}}", src)
    return [int(v) for v in m.group(1).split(",")]


def test_index_tables_are_the_references():
    src = "".join(open(os.path.join(ROOT, "mhentropy_amd", "csrc", f)).read() for f in ("mano_joint_pass.h", "mano.hip"))
    assert _c_array(src, "kJointReorder") == list(mano_ref.JOINT_REORDER)
    assert _c_array(src, "kFreihand2Rhd") == list(mano_ref.FREIHAND2RHD)
    assert tuple(mano_pack.TIP_VERTS_RIGHT) == mano_ref.TIP_VERTS_RIGHT
    # centring joint: index 9 of the reordered list (manolayer.py:262-266)
    assert int(re.search(r"kCenterPre = (\d+);", src).group(1)) == mano_ref.JOINT_REORDER[9]
    # wrapper re-regression map and tip vertices (ManoLayer.py:112-126)
    jm = _c_array(src, "kWrapJointMap")
    assert {i: v for i, v in enumerate(jm)} == mano_ref.WRAPPER_JOINT_MAP
    assert _c_array(src, "kWrapTipVert") == [mano_ref.WRAPPER_TIP_VERTS[k] for k in (4, 8, 12, 16, 20)]


@pytest.mark.skipif(not os.path.isdir("/root/reference/hand"), reason="the reference tree only exists in the build container")
def test_index_tables_against_the_reference_source_text():
    """the same tables parsed straight out of the reference's source text (reading it is study, nothing is imported):
    hand/manopth/manolayer.py:197-199,228,251,260, hand/utils.py:15, hand/ManoLayer.py:112-127"""
    ml = open("/root/reference/hand/manopth/manolayer.py").read()
    lists = [[int(v) for v in m.split(",")] for m in re.findall(r"\[((?:\s*\d+\s*,)+\s*\d+\s*)\]", ml)]
    assert [1, 4, 7, 10, 13] in lists and [2, 5, 8, 11, 14] in lists and [3, 6, 9, 12, 15] in lists
    assert [0, 1, 6, 11, 2, 7, 12, 3, 8, 13, 4, 9, 14, 5, 10, 15] in lists                   # reorder_idxs
    assert list(mano_ref.TIP_VERTS_RIGHT) in lists                                            # [745, 317, 444, 556, 673]
    assert list(mano_ref.JOINT_REORDER) in lists
    ut = open("/root/reference/hand/utils.py").read()
    m = re.search(r"FreiHand2RHD_skeidx\s*=\s*\[([^\]]*)\]", ut)
    assert [int(v) for v in m.group(1).split(",")] == list(mano_ref.FREIHAND2RHD)
    wl = open("/root/reference/hand/ManoLayer.py").read()
    for k, v in mano_ref.WRAPPER_TIP_VERTS.items():
        assert re.search(rf"\b{v}\b", wl), (k, v)


def test_flow_pack_roundtrip():
    """every weight appears exactly once in the packed stream at the documented position"""
    from mhentropy_amd import ops
    rng = np.random.default_rng(0)
    h, d = 64, 45
    w0, w1, w2 = rng.normal(size=(h, d)), rng.normal(size=(h, h)), rng.normal(size=(d, h))
    p = ops.flow_pack_net(w0.astype(np.float32), w1.astype(np.float32), w2.astype(np.float32))
    assert p.shape == (3 * 4096,)
    nz = p[p != 0]
    assert nz.size == w0.size + w1.size + w2.size
    assert np.isclose(np.sort(nz), np.sort(np.concatenate([w0.ravel(), w1.ravel(), w2.ravel()]).astype(np.float32))).all()
    # block (To,Tk) of layer 0 lives at block f = Tk*NT + To; lane 16q+n holds W[16To+n][16Tk+4q+i]
    NT, To, Tk, q, n, i = 4, 2, 1, 3, 5, 2
    assert p[(Tk * NT + To) * 256 + (16 * q + n) * 4 + i] == np.float32(w0[16 * To + n, 16 * Tk + 4 * q + i])


def test_state_dict_keys_match_reference_layout():
    """key names a reference checkpoint (ent_ho3d.pth) carries for encoderRGB"""
    from mhentropy_amd import harness
    model = harness.build_mhent(backbone="resnet18", h_dims=(64, 64), num_steps=2, tables=synth.mano_tables(0))
    keys = set(model.state_dict().keys())
    for k in ("feat_extractor.res.conv1.weight", "feat_extractor.res.bn1.running_mean",
              "feat_extractor.res.layer2.0.downsample.0.weight", "feat_extractor.res.layer4.1.bn2.num_batches_tracked",
              "feat_extractor.l1.0.weight", "feat_extractor.l2.0.bias", "q_z_giv_i.mask", "q_z_giv_i.s.0.l.0.weight",
              "q_z_giv_i.t.3.c.1.bias", "det_head.0.weight", "det_head.2.bias", "mano_dec.mano_layer.th_posedirs",
              "mano_dec.mano_layer.th_selected_comps", "mano_dec.mano_layer.th_faces"):
        assert k in keys, k
    sd = {"q_z_giv_i." + k: torch.as_tensor(v) for k, v in synth.flow_state(1, 45, 512, (64, 64), 2).items()}
    sd.update({k: torch.as_tensor(v) for k, v in synth.head_state(1, 512, 512, 16).items()})
    sd.update({"feat_extractor.res." + k: torch.as_tensor(v) for k, v in synth.resnet_state(1, "resnet18").items()})
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    assert all(k.startswith("mano_dec.") for k in missing)


def test_launcher_picks_the_documented_kernel_variants_at_c2():
    """host logic only (mhe_conv_tile_mode runs no kernel): the variant table of DESIGN.md section 4 at the bench workload's shapes
    (B = 256, ResNet-50 at 256x256, bf16) - streaming kernels for the write-bound layers, transfer-wave kernels for the wide ones"""
    import torch
    from mhentropy_amd import ops
    bf, B = torch.bfloat16, 256
    pick = lambda H, cin, cout, k, stride=1, mode=0: ops.conv_tile_choice(B, H, H, cin, cout, k, stride, k // 2, bf, mode)
    assert pick(64, 64, 64, 3) == 9 and pick(64, 64, 64, 3, mode=1) == 9          # layer1 conv2: row-streaming 3x3, also with BatchNorm on load
    assert pick(64, 64, 256, 1) == 8 and pick(64, 64, 256, 1, mode=1) == 8        # layer1 conv3 / shortcut: streaming 1x1
    assert pick(32, 128, 512, 1, mode=1) == 8                                     # layer2 conv3
    assert pick(16, 256, 1024, 1) == 11 and pick(16, 256, 1024, 1, mode=1) == 11  # layer3 conv3: resident slab + transfer waves
    assert pick(8, 512, 2048, 1) == 7                                             # layer4 conv3 stays on the phase-pipelined kernel
    assert pick(16, 1024, 256, 1, mode=2) == 10 and pick(8, 2048, 512, 1, mode=2) == 10      # residual tails of layer3 / layer4
    assert pick(32, 512, 256, 1, mode=2) == 10 and pick(16, 1024, 512, 1, mode=2) == 10      # ... and the transitions into them
    assert pick(64, 256, 64, 1, mode=2) == 0 and pick(32, 512, 128, 1, mode=2) == 1          # layer1 / layer2 tails: register-staged tiles
    assert pick(16, 256, 256, 3) == 7                                             # 3x3: phase-pipelined 256x256 where its tiles fill the chip
    assert pick(8, 512, 512, 3) == 13 and pick(16, 512, 512, 3, stride=2) == 13   # ... layer4 (16k pixels): the same kernel on 256x128 tiles
    assert pick(32, 128, 128, 3) == 1                                             # layer2 (K = 1152): register-staged 128x128, two workgroups per CU
    # small batches fall back to the tiled kernels (the streaming / transfer-wave kernels want whole persistent workgroups)
    assert ops.conv_tile_choice(4, 16, 16, 256, 1024, 1, 1, 0, bf, 0) not in (8, 9, 10, 11)


def test_product_refuses_cpu_tensors():
    from mhentropy_amd import ops
    with pytest.raises(_lib.MheError):
        ops.linear(torch.zeros(4, 32), torch.zeros(8, 32))


def test_fragment_major_order_is_the_headers_formula():
    """include/mhe.h (mhe_flow_reverse_chain_bf16): element [out][k] of an operand lies at
    ((out / 16 * (K / 32) + k / 32) * 64 + (k % 32 / 8) * 16 + out % 16) * 8 + k % 8"""
    import torch
    from mhentropy_amd import ops
    rows, K = 64, 128
    idx = torch.arange(rows * K).view(rows, K)
    flat = ops.mfma_fragment_major(idx).reshape(-1)
    o, k = torch.meshgrid(torch.arange(rows), torch.arange(K), indexing="ij")
    pos = ((o // 16 * (K // 32) + k // 32) * 64 + (k % 32 // 8) * 16 + o % 16) * 8 + k % 8
    assert torch.equal(flat[pos.reshape(-1)], idx.reshape(-1))


def test_round3_entries_refuse_what_they_do_not_support():
    """argument checks of the round-3 C entries run before anything touches the GPU: unsupported geometries and null operands come back
    as MHE_ERR_ARG with a message, never as a launch"""
    L = _lib.lib()
    # hidden 512, whole images (round 5: any hypothesis count per image in the forward-only form)
    assert L.mhe_flow_couplings_frag_supported(64 * 3, 3, 45, 512, 12) == 1 and L.mhe_flow_couplings_frag_supported(40 * 3, 3, 45, 512, 12) == 1
    assert L.mhe_flow_couplings_frag_supported(40 * 3 + 1, 3, 45, 512, 12) == 0 and L.mhe_flow_couplings_frag_supported(64 * 3, 3, 45, 256, 12) == 0
    assert L.mhe_flow_couplings_frag_supported(64 * 3, 3, 45, 512, 40) == 0
    assert L.mhe_flow_reverse_chain_supported(64 * 3, 3, 45, 512, 12) == 1 and L.mhe_flow_reverse_chain_supported(128 * 3, 3, 45, 512, 12) == 0
    null = None
    rc = L.mhe_flow_couplings_frag_bf16(null, null, null, 0, null, null, null, 0, null, null, null, null, null, null, null, null, 192, 3, 45, 512, 12, 0, null)
    assert rc != 0 and b"null pointer" in L.mhe_last_error()
    rc = L.mhe_flow_reverse_chain_bf16(null, null, null, 0.0, null, null, null, null, null, null, 0, null, null, null, null, null, 0, null, null,
                                       192, 3, 45, 512, 12, null)
    assert rc != 0 and b"null pointer" in L.mhe_last_error()
    rc = L.mhe_conv3_bn_fold(null, null, null, null, null, null, 1024.0, null, null, null, null, 256, null, 64, null, null, 256, 64, null)
    assert rc != 0 and b"null pointer" in L.mhe_last_error()
    rc = L.mhe_conv3x3_halo_nhwc(256, 32, 32, 128, 128, null, null, null, null, null, 0, null, null, null, null, null, null, null, null)
    assert rc != 0 and b"null pointer" in L.mhe_last_error()
    rc = L.mhe_conv3x3_halo_pack_bf16(null, null, 128, 128, null)
    assert rc != 0 and b"null pointer" in L.mhe_last_error()
    assert L.mhe_conv3x3_halo_supported(256, 32, 32, 128, 128) == 1 and L.mhe_conv3x3_halo_supported(256, 8, 8, 512, 512) == 0
    rc = L.mhe_pack_transpose_bf16(null, 0, null, null, 4, 4, null)
    assert rc != 0 and b"bad arguments" in L.mhe_last_error()


def test_committed_counter_summaries_describe_the_committed_kernels():
    """bench.py's `roofline.traffic` comes from profiles/<round>_pmc_traffic.json and is reported only while the summary's
    `source_sha1` equals the sha1 of csrc/*.hip + csrc/*.h; a kernel edit after the counter passes would silently turn the field into
    null.  This test makes that a failure instead: re-run tools/refresh_profiles_r05.sh after touching a kernel."""
    sys.path.insert(0, ROOT)
    import bench
    from tools.pmc_traffic import kernel_sources_sha1
    sha = kernel_sources_sha1()
    for name in (bench.PMC_TRAFFIC, bench.PMC_TRAFFIC_TRAIN):
        path = os.path.join(ROOT, "profiles", name)
        assert os.path.exists(path), path
        d = json.load(open(path))
        assert d["source_sha1"] == sha, f"{name} was collected on other kernel sources ({d['source_sha1'][:10]} vs {sha[:10]}): refresh it"
        assert d["kernels"] and all(v["hbm_bytes_per_launch"] >= 0 for v in d["kernels"].values())


def test_switch_table_matches_the_source():
    """mhentropy_amd/switches.py is the ONE table of run-time switches: every MHE_* environment variable the host code or the kernels'
    launchers read is listed there with the default the source uses (so "no MHE_* set" IS the benched product path), and the table lists
    nothing the source does not read"""
    import glob
    import re
    from mhentropy_amd import switches
    found = {}
    pats = [re.compile(r'environ\.get\("(MHE_[A-Z0-9_]+)"(?:,\s*"([^"]*)")?\)'), re.compile(r'getenv\("(MHE_[A-Z0-9_]+)"\)')]
    files = glob.glob(os.path.join(ROOT, "mhentropy_amd", "*.py")) + glob.glob(os.path.join(ROOT, "mhentropy_amd", "csrc", "*.hip")) + \
        glob.glob(os.path.join(ROOT, "mhentropy_amd", "csrc", "*.h")) + [os.path.join(ROOT, "bench.py")]
    for f in files:
        if f.endswith("switches.py") or f.endswith("build.py"):
            continue
        src = open(f).read()
        for pat in pats:
            for m in pat.finditer(src):
                found.setdefault(m.group(1), set())
                if pat is pats[0] and m.group(2) is not None:
                    found[m.group(1)].add(m.group(2))
        # C side: `getenv("X") ? atoi(getenv("X")) : D`
        for m in re.finditer(r'getenv\("(MHE_[A-Z0-9_]+)"\)\s*\?\s*ato[il]\(getenv\("MHE_[A-Z0-9_]+"\)\)\s*:\s*(-?\d+)', src):
            found[m.group(1)].add(m.group(2))
    assert set(found) == set(switches.SWITCHES), (sorted(set(found) - set(switches.SWITCHES)), sorted(set(switches.SWITCHES) - set(found)))
    for k, defaults in found.items():
        assert all(d == switches.SWITCHES[k][0] for d in defaults), (k, defaults, switches.SWITCHES[k][0])
    # DESIGN.md prints this very table
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert switches.markdown_table().strip() in design, "DESIGN.md's switch table is not mhentropy_amd.switches.markdown_table()"
    assert switches.non_default({}) == {} and switches.non_default({"MHE_CONV_HALO": "1"}) == {} and switches.non_default({"MHE_CONV_HALO": "0"}) == {"MHE_CONV_HALO": "0"}
