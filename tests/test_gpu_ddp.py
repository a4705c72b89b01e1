"""Two ranks sharing the one GPU of the test box, talking over gloo: the train step's gradient exchange (bucketed
all-reduce issued while the reverse pass is still running) gives every rank the mean gradient of the
two image shards and identical parameters after the step.  (The driver's multi-GPU runs use RCCL, one GPU per rank.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

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
           "--master-port", "29633", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-3000:]
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    # all-reduced gradient == mean of the shards' gradients (recomputed in a second run: atomics order differs, so fp32 noise)
    assert max(r["err"] for r in recs) < 2e-3, recs
    assert recs[0]["psum"] == recs[1]["psum"], recs                # replicas stay bit-identical after clip + Adam
    assert abs(recs[0]["sq"] - recs[1]["sq"]) == 0.0
    # attach() + step(): the gradient is averaged once (by backward(), for torch's optimizer), not again by the optimizer kernel
    assert max(r["bridge_avg_err"] for r in recs) < 2e-3, recs
    assert all(r["bridge_scale"] == 1.0 and r["plain_scale"] == 0.5 for r in recs), recs
