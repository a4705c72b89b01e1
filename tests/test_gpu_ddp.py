"""Two ranks sharing the one GPU of the test box, talking over gloo: the train step's gradient exchange (bucketed
all-reduce issued while the reverse pass is still running) gives every rank the mean gradient of the
two image shards and identical parameters after the step.  (The driver's multi-GPU runs use RCCL, one GPU per rank.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, free_port

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MHE_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MHE_ROOT"], "tests"))
import torch
from mhentropy_amd import dist as mdist, synth
from mhentropy_amd.train import TrainStep
from test_gpu_train import _model_and_state
rank, _, world, dist = mdist.init("gloo")
torch.cuda.set_device(0)
B, N = 3, 4
def shard(r):
    xn, yn = synth.batch(40 + r, B, image_size=96)
    return torch.as_tensor(xn).cuda(), {k: torch.as_tensor(v).cuda() for k, v in yn.items()}, torch.as_tensor(synth.noise(40 + r, N * B)).cuda()
# expected: mean over the two shards of the single-process gradients (same initial weights on every rank)
model, _ = _model_and_state("resnet18", 64, 2)
solo = TrainStep(model)
want = None
for r in range(world):
    x, y, z0 = shard(r)
    solo.forward_backward(x, y, noise=z0, N=N)
    want = solo.G.clone() if want is None else want + solo.G
want /= world
model2, _ = _model_and_state("resnet18", 64, 2)
ts = TrainStep(model2, dist=dist)
x, y, z0 = shard(rank)
ts.forward_backward(x, y, noise=z0, N=N)
assert len(ts._works) == 4, "four gradient buckets in flight"
ts.finish_allreduce()
err = ((ts.G / world - want).abs().max() / want.abs().max()).item()
ts.forward_backward(x, y, noise=z0, N=N)
ts.optimizer_step()
psum = float(ts.P.double().sum())
# attach() bridge + fused step(): backward() already averages G for torch's optimizer; optimizer_step must not divide by world again
model3, _ = _model_and_state("resnet18", 64, 2)
tb = TrainStep(model3, dist=dist, lr=0.0).attach()
tb.forward_backward(x, y, noise=z0, N=N)
avg_err = ((tb.G - want).abs().max() / want.abs().max()).item()        # .grad holds the mean over ranks for torch's optimizer
tb.optimizer_step()
with open(os.path.join(os.environ["MHE_OUT"], f"rank{rank}.json"), "w") as fh:
    json.dump({"err": err, "psum": psum, "sq": float(ts.sq), "bridge_avg_err": avg_err, "bridge_scale": tb.last_grad_scale, "plain_scale": ts.last_grad_scale}, fh)
dist.destroy_process_group()
'''


def test_two_ranks_average_their_gradients(gpu_lib, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MHE_ROOT=ROOT, MHE_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-3000:]
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    # all-reduced gradient == mean of the shards' gradients (recomputed in a second run; bound kept from the rounds with order-dependent f32 atomics - the sums are order-independent since round 4)
    assert max(r["err"] for r in recs) < 2e-3, recs
    assert recs[0]["psum"] == recs[1]["psum"], recs                # replicas stay bit-identical after clip + Adam
    assert abs(recs[0]["sq"] - recs[1]["sq"]) == 0.0
    # attach() + step(): the gradient is averaged once (by backward(), for torch's optimizer), not again by the optimizer kernel
    assert max(r["bridge_avg_err"] for r in recs) < 2e-3, recs
    assert all(r["bridge_scale"] == 1.0 and r["plain_scale"] == 0.5 for r in recs), recs


X2_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MHE_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MHE_ROOT"], "tests"))
import torch
from mhentropy_amd import dist as mdist, synth
from mhentropy_amd.train import TrainStep
from test_gpu_train import _model_and_state
rank, _, world, dist = mdist.init("gloo")
torch.cuda.set_device(0)
B, N = 3, 4
xn, yn = synth.batch(60 + rank, B, image_size=96)
x, y = torch.as_tensor(xn).cuda(), {k: torch.as_tensor(v).cuda() for k, v in yn.items()}
z0 = torch.as_tensor(synth.noise(60 + rank, N * B)).cuda()
res = {}
for name, sharded in (("images", False), ("hypotheses", True)):
    model, _ = _model_and_state("resnet18", 64, 2)
    ts = TrainStep(model, dist=dist, shard_hypotheses=sharded)
    out = ts.forward_backward(x, y, noise=z0, N=N)
    ts.finish_allreduce()
    res[name] = (out["log_p"].clone(), out["h_q_z_giv_i"].clone(), ts.G.clone() / world, ts.tape["g_feat"].clone())
a, b = res["images"], res["hypotheses"]
rel = lambda u, v: float((u - v).abs().max() / (v.abs().max() + 1e-30))
with open(os.path.join(os.environ["MHE_OUT"], f"x2_rank{rank}.json"), "w") as fh:
    json.dump({"log_p": rel(b[0], a[0]), "h": rel(b[1], a[1]), "grad": rel(b[2], a[2]), "g_feat": rel(b[3], a[3])}, fh)
dist.destroy_process_group()
'''


def test_hypothesis_sharded_train_step_equals_the_image_sharded_one(gpu_lib, tmp_path):
    """TrainStep(shard_hypotheses=True) on two ranks (gloo, one GPU): same per-image loss terms, same dL/dfeat for the rank's own
    images and the same averaged flat gradient as the plain image-sharded step (north_star "optionally
    hypotheses")"""
    script = tmp_path / "x2_worker.py"
    script.write_text(X2_WORKER)
    env = dict(os.environ, MHE_ROOT=ROOT, MHE_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-3000:]
    for r in range(2):
        rec = json.load(open(tmp_path / f"x2_rank{r}.json"))
        assert rec["log_p"] < 1e-5 and rec["h"] < 1e-5, rec
        # two runs of the reverse pass (the bound dates from the order-dependent f32 atomics of rounds 1-3: see test_two_ranks_average_their_gradients)
        assert rec["g_feat"] < 2e-3 and rec["grad"] < 2e-3, rec


GRAPH_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MHE_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MHE_ROOT"], "tests"))
import torch
from mhentropy_amd import dist as mdist, synth
from mhentropy_amd.train import TrainStep, GraphedStep
from test_gpu_train import _model_and_state
rank, _, world, dist = mdist.init("gloo")
torch.cuda.set_device(0)
B, N = 3, 4
xn, yn = synth.batch(70 + rank, B, image_size=96)
x, y = torch.as_tensor(xn).cuda(), {k: torch.as_tensor(v).cuda() for k, v in yn.items()}
z0 = torch.as_tensor(synth.noise(70 + rank, N * B)).cuda()
model, _ = _model_and_state("resnet18", 64, 2)
ts = TrainStep(model, dist=dist, lr=0.0)                   # lr 0: every step sees the same parameters
out = ts.step(x, y, noise=z0, N=N)
G_eager, loss_eager = ts.G.clone(), float(out["total"])
gs = GraphedStep(ts, x, y, noise=z0, N=N)
res = []
for _ in range(2):
    o = gs.replay(); torch.cuda.synchronize()
    res.append((float((ts.G - G_eager).abs().max() / G_eager.abs().max()), float(o["total"])))
model2, _ = _model_and_state("resnet18", 64, 2)
ts2 = TrainStep(model2, dist=dist, lr=1e-3)
gs2 = GraphedStep(ts2, x, y, noise=z0, N=N)                # capture itself takes one eager step
gs2.replay(); gs2.replay(); torch.cuda.synchronize()
with open(os.path.join(os.environ["MHE_OUT"], f"graph_rank{rank}.json"), "w") as fh:
    json.dump({"graphs": len(gs.graphs), "actions": [a[0] for a in gs.actions], "err": [r[0] for r in res], "loss": [r[1] for r in res],
               "loss_eager": loss_eager, "psum": float(ts2.P.double().sum()), "steps": int(ts2.step_t.item()) if hasattr(ts2.step_t, "item") else -1}, fh)
dist.destroy_process_group()
'''


def test_graphed_step_cut_at_the_gradient_buckets(gpu_lib, tmp_path):
    """GraphedStep under data parallelism (two ranks, gloo, one GPU): the step replays from six HIP graphs with the four bucket
    all-reduces issued between them; same all-reduced gradient and loss as the eager step, replicas bit-identical after Adam"""
    script = tmp_path / "graph_worker.py"
    script.write_text(GRAPH_WORKER)
    env = dict(os.environ, MHE_ROOT=ROOT, MHE_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-3000:]
    recs = [json.load(open(tmp_path / f"graph_rank{r}.json")) for r in range(2)]
    for r in recs:
        assert r["graphs"] == 6 and r["actions"] == ["allreduce"] * 4 + ["wait"], r
        assert max(r["err"]) < 2e-3, r                      # (bound from rounds 1-3: f32 atomics; deterministic since round 4)
        assert all(abs(l - r["loss_eager"]) <= 1e-5 * abs(r["loss_eager"]) for l in r["loss"]), r
    assert recs[0]["psum"] == recs[1]["psum"], recs
    assert recs[0]["steps"] == recs[1]["steps"] == 3, recs   # the eager warm-up step inside GraphedStep + two replays (capture executes nothing)
