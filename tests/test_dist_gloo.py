"""CPU, world_size 2 over gloo: the multi-process bookkeeping bench.py relies on (rank shards that
tile the global batch, barrier-bracketed max-over-ranks timing, scalar averaging)."""
import os
import subprocess
import sys

from conftest import ROOT
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
           "--master-port", "29611", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert (recs[0]["lo"], recs[0]["hi"], recs[1]["lo"], recs[1]["hi"]) == (0, 4, 4, 7)
    assert recs[0]["calls"] == recs[1]["calls"] == 3                       # exactly K timed steps per rank
    assert abs(recs[0]["dt"] - recs[1]["dt"]) < 1e-9                       # everyone reports the max over ranks
    assert recs[0]["dt"] >= 3 * 0.04 - 1e-3                                # ... which is the slow rank's time
    assert recs[0]["avg"] == recs[1]["avg"] == {"loss": 0.5, "n": 3.5}
    assert recs[0]["sum_x"] != recs[1]["sum_x"]                            # different shards


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
