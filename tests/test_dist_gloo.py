"""CPU, world_size 2 over gloo: the multi-process bookkeeping bench.py relies on (rank shards that
tile the global batch, barrier-bracketed max-over-ranks timing, scalar averaging)."""
import os
import subprocess
import sys

from conftest import ROOT, free_port
from mhentropy_amd import dist as mdist

WORKER = r'''
import os, sys, time, json
sys.path.insert(0, os.environ["MHE_ROOT"])
import torch
from mhentropy_amd import dist as mdist, synth
rank, local_rank, world, dist = mdist.init("gloo")
assert world == 2 and dist is not None
lo, hi = mdist.shard_range(7, rank, world)
x, y = synth.batch(100 + rank, hi - lo, image_size=8)            # rank-private synthetic shard
calls = []
def step():
    calls.append(1); time.sleep(0.02 * (rank + 1))                # rank 1 is slower
dt = mdist.timed_region(step, 3, dist, sync=lambda: None)
avg = mdist.reduce_mean_scalars({"loss": float(rank), "n": float(hi - lo)}, dist)
g = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # the train step's flat gradient exchange, 4 buckets
mdist.allreduce_gradients(g, dist, bucket_elems=300)
assert torch.equal(g, torch.arange(1000, dtype=torch.float32) * 3), "bucketed gradient all-reduce"
with open(os.path.join(os.environ["MHE_OUT"], f"rank{rank}.json"), "w") as fh:
    json.dump({"rank": rank, "lo": lo, "hi": hi, "dt": dt, "calls": len(calls), "avg": avg, "sum_x": float(x.sum())}, fh)
dist.destroy_process_group()
'''


def test_shard_ranges_tile_the_batch():
    for gb in (1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [mdist.shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_over_gloo(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MHE_ROOT=ROOT, MHE_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert (recs[0]["lo"], recs[0]["hi"], recs[1]["lo"], recs[1]["hi"]) == (0, 4, 4, 7)
    assert recs[0]["calls"] == recs[1]["calls"] == 3                       # exactly K timed steps per rank
    assert abs(recs[0]["dt"] - recs[1]["dt"]) < 1e-9                       # everyone reports the max over ranks
    assert recs[0]["dt"] >= 3 * 0.04 - 1e-3                                # ... which is the slow rank's time
    assert recs[0]["avg"] == recs[1]["avg"] == {"loss": 0.5, "n": 3.5}
    assert recs[0]["sum_x"] != recs[1]["sum_x"]                            # different shards


X2_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MHE_ROOT"])
import torch
from mhentropy_amd import dist as mdist
rank, _, world, dist = mdist.init("gloo")
torch.manual_seed(0)
B, K, F = 3, 6, 8                      # images per rank, hypotheses per image, feature width
W = torch.randn(F, 5, dtype=torch.float64)                                   # "flow + decoder" parameters (replicated)
feat_all = torch.randn(world * B, F, dtype=torch.float64)
noise_all = torch.randn(K, world * B, 5, dtype=torch.float64)               # sample-major over the global batch
def rows_loss(feat, noise):          # per-row log-density of a toy conditional model: rows (n, b)
    return -((feat @ W)[None] + noise).pow(2).sum(-1) + torch.sin(feat.sum(-1))[None]
# unsharded reference on the global batch: per-image mean over K, loss = mean over images
fa = feat_all.clone().requires_grad_(True); Wa = W.clone().requires_grad_(True)
W_saved = W; W = Wa
per_img = rows_loss(fa, noise_all).mean(0)
(-per_img.mean()).backward()
W = W_saved
# sharded: own images' features -> gather; own hypothesis slice for ALL images; partial sums -> all-reduce; grad -> reduce-scatter
hs = mdist.HypothesisShards(dist, K)
own = slice(rank * B, (rank + 1) * B)
feat_own = feat_all[own].clone()
noise_own = noise_all[:, own].reshape(K * B, 5)                              # what a rank holds: sample-major over ITS images
g = hs.gather_rows(feat_own).clone().requires_grad_(True)
Wl = W.clone().requires_grad_(True)
W = Wl
lo, hi = hs.hypotheses()
nz = hs.gather_hypothesis_rows(noise_own, B).reshape(hi - lo, world * B, 5)
assert torch.equal(nz, noise_all[lo:hi]), "the rank's hypotheses of all images, sample-major"
rows = rows_loss(g, nz)
part = rows.sum(0) / K
tot = hs.reduce_images(part.detach().clone())
# reverse: d(-mean_b per_img)/d rows, with the per-rank convention -1/B_own followed by / world
(part * (-1.0 / B)).sum().backward()
g_own = hs.scatter_grad(g.grad.clone()) / world
gW = Wl.grad.clone(); dist.all_reduce(gW); gW /= world
rec = {"per_img_err": float((tot - per_img.detach()).abs().max()), "gfeat_err": float((g_own - fa.grad[own]).abs().max()),
       "gW_err": float((gW - Wa.grad).abs().max()), "lo": lo, "hi": hi}
with open(os.path.join(os.environ["MHE_OUT"], f"x2_rank{rank}.json"), "w") as fh:
    json.dump(rec, fh)
dist.destroy_process_group()
'''


def test_hypothesis_sharded_exchange_equals_the_unsharded_loss_and_gradient(tmp_path):
    """dist.HypothesisShards over gloo, world 2 (SURVEY.md section 8e optional hypothesis sharding): gathering the features,
    evaluating a slice of the hypotheses for all images, all-reducing the per-image sums and reduce-scattering dL/dfeat gives
    the unsharded per-image values, feature gradients and (after the ordinary gradient all-reduce) parameter gradients"""
    import json
    script = tmp_path / "x2_worker.py"
    script.write_text(X2_WORKER)
    env = dict(os.environ, MHE_ROOT=ROOT, MHE_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.load(open(tmp_path / f"x2_rank{r}.json")) for r in range(2)]
    assert [(r["lo"], r["hi"]) for r in recs] == [(0, 3), (3, 6)]
    for r in recs:
        assert r["per_img_err"] < 1e-12 and r["gfeat_err"] < 1e-12 and r["gW_err"] < 1e-12, r


def test_trainer_shell_helpers(tmp_path):
    """row f3: AverageMeter quirk, MultiStepLR schedule, checkpoint container (reference hand/utils.py:75-91,
    hand/CrossModalHand.py:202,573-602)"""
    import torch
    from mhentropy_amd import harness
    m = harness.AverageMeter()
    for v in (2.0, 0.0, 4.0):
        m.update(v, n=8)
    assert (m.count, m.avg, m.val) == (2, 3.0, 4.0)
    assert [harness.multistep_lr(2e-4, e) for e in (0, 149, 150, 249, 250, 400)] == [2e-4, 2e-4, 2e-5, 2e-5, 2e-4 * 0.1 ** 2, 2e-4 * 0.1 ** 2]
    # ... and against torch's own scheduler, stepped once per epoch like the reference does
    opt_ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=2e-4)
    sched_ref = torch.optim.lr_scheduler.MultiStepLR(opt_ref, milestones=[150, 250], gamma=0.1)
    for epoch in range(300):
        assert abs(harness.multistep_lr(2e-4, epoch) - opt_ref.param_groups[0]["lr"]) < 1e-12, epoch
        opt_ref.step(); sched_ref.step()
    lin = torch.nn.Linear(3, 2)
    harness.save_model(tmp_path / "ck.pth", lin)
    ck = torch.load(tmp_path / "ck.pth")
    assert set(ck) == {"decoderPose", "encoderRGB"} and set(ck["encoderRGB"]) == {"weight", "bias"}
    lin2 = torch.nn.Linear(3, 2)
    harness.load_model(tmp_path / "ck.pth", lin2)
    assert torch.equal(lin2.weight, lin.weight)


def test_config_ingestion_matches_the_reference_tree(tmp_path):
    """row f3: a YAML file in the reference's schema -> (special_cfg, common_cfg) as hand/CrossModalHand.py:53-85 builds them;
    with the reference's own configs/ho3d.yaml (read as data where the tree exists) this is the shipped model"""
    import os
    import pytest
    from mhentropy_amd import harness
    cfg = harness.load_config(os.path.join(ROOT, "configs", "c2_bench.yaml"))
    special, common = harness.mhent_cfgs_from_config(cfg)
    want_s, want_c = harness.mhent_cfgs(backbone="resnet50", h_dims=(512, 512), num_steps=6)
    assert special == want_s and common == want_c
    assert cfg.training.batch_size == 256 and cfg.training.epochs == 80            # file value, default value
    with pytest.raises(KeyError):
        bad = tmp_path / "bad.yaml"
        bad.write_text("dataset:\n  no_such_key: 1\n")
        harness.load_config(bad)
    ref = "/root/reference/hand/configs/ho3d.yaml"
    if os.path.exists(ref):
        rc = harness.load_config(ref)
        rs, rcom = harness.mhent_cfgs_from_config(rc)
        assert rs == want_s and {k: v for k, v in rcom.items()} == want_c
        assert (rc.training.lr, list(rc.training.milestones), rc.training.batch_size, rc.training.test_samples) == (2e-4, [150, 250], 64, 200)
        assert rc.training.criterion == "MHEntLoss" and rc.network.regressor == "realnvp"


def test_checkpoint_container_round_trip_of_the_full_model(tmp_path):
    """row f3: harness.save_model / load_model on the whole MHEnt state_dict in the reference's container
    (hand/CrossModalHand.py:573-602: {'decoderPose', 'encoderRGB'}), every key family of SURVEY.md section 8b present"""
    import torch
    from mhentropy_amd import harness, synth
    m = harness.build_mhent(backbone="resnet18", h_dims=(64, 64), num_steps=2, tables=synth.mano_tables(0))
    with torch.no_grad():
        for p_ in m.parameters():
            p_.add_(torch.randn_like(p_) * 0.01)
    harness.save_model(tmp_path / "ent.pth", m)
    ck = torch.load(tmp_path / "ent.pth")
    assert set(ck) == {"decoderPose", "encoderRGB"} and ck["decoderPose"] == {}
    keys = set(ck["encoderRGB"])
    for fam in ("feat_extractor.res.conv1.weight", "feat_extractor.res.layer4.1.bn2.running_var", "feat_extractor.l1.0.weight",
                "feat_extractor.l2.0.bias", "q_z_giv_i.mask", "q_z_giv_i.s.3.l.2.weight", "q_z_giv_i.t.0.c.1.bias", "det_head.0.weight",
                "det_head.2.bias", "mano_dec.mano_layer.th_posedirs", "mano_dec.mano_layer.th_faces"):
        assert fam in keys, fam
    m2 = harness.build_mhent(backbone="resnet18", h_dims=(64, 64), num_steps=2, tables=synth.mano_tables(0))
    harness.load_model(tmp_path / "ent.pth", m2)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_scalar_log_uses_the_reference_tags(tmp_path):
    import json
    import torch
    from mhentropy_amd import harness
    log = harness.ScalarLog(tmp_path / "scalars.jsonl")
    log.iteration(3, {"neg_log_p": torch.tensor([1.0, 3.0])}, {"eucLoss_3d_rgb_sample": torch.tensor([0.5])},
                  {"th_norm": torch.tensor([2.0]), "bt_norm": torch.tensor([0.1])})
    log.epoch(4, 2.0, 0.012)
    log.close()
    recs = [json.loads(l) for l in open(tmp_path / "scalars.jsonl")]
    tags = {r["tag"]: r for r in recs}
    assert tags["loss_it/neg_log_p"]["value"] == 2.0 and tags["param/theta_norm"]["value"] == 2.0
    assert tags["loss_avg/loss_total"]["step"] == 4 and abs(tags["metric_train/eval_3d_rgb"]["value"] - 12.0) < 1e-9


XCHG_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MHE_ROOT"])
import torch
from mhentropy_amd import dist as mdist
rank, _, world, dist = mdist.init("gloo")
n = 1003                                       # not a multiple of 8 * world: the chunks are padded
gs = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) * (1.0 + r) for r in range(world)]
rec = {}
for mode in ("f32", "bf16"):
    x = mdist.GradExchange(dist, mode=mode)
    g = gs[rank].clone()
    x.start(g, key=0)
    x.finish()
    if mode == "f32":
        want = sum(gs)
    else:                                      # bf16 on the wire, f32 accumulation in rank order, one bf16 rounding of the sum
        want = sum(t.bfloat16().float() for t in gs).bfloat16().float()
    rec[mode] = {"equal": bool(torch.equal(g, want)), "err": float((g - want).abs().max()), "sum": float(g.double().sum())}
json.dump(rec, open(os.path.join(os.environ["MHE_OUT"], f"xchg{rank}.json"), "w"))
dist.destroy_process_group()
'''


def test_gradient_exchange_modes_over_gloo(tmp_path):
    """dist.GradExchange on two ranks: the f32 all-reduce, and the bf16 exchange (all-to-all of bf16 chunks, local f32 sum in rank order, bf16
    all-gather: half the bytes of the f32 ring with f32 accumulation) - every rank ends with the same bits, equal to the value computed
    locally from both ranks' gradients"""
    import json
    script = tmp_path / "xchg.py"
    script.write_text(XCHG_WORKER)
    env = dict(os.environ, MHE_ROOT=ROOT, MHE_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.load(open(tmp_path / f"xchg{r}.json")) for r in range(2)]
    for mode in ("f32", "bf16"):
        assert recs[0][mode]["equal"] and recs[1][mode]["equal"], (mode, recs)
        assert recs[0][mode]["sum"] == recs[1][mode]["sum"]                 # replicas hold the same bits
