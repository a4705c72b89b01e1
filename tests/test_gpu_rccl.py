"""The RCCL code path on the hardware the test box has: ONE rank, backend "nccl" (= RCCL on ROCm), MHE_DIST_FORCE=1 so that every
collective the N-GPU run issues is issued here too.  A group of one proves nothing about xGMI, but it executes
`init_process_group("nccl", device_id=...)`, the communicator's stream ordering against the compute / capture streams,
`all_reduce(async_op=True)` between the six HIP graphs of GraphedStep (`capture_error_mode="thread_local"` beside a live communicator),
`all_gather_into_tensor` and `reduce_scatter_tensor` of the hypothesis-sharded exchange - before the first 8-GPU run does
(the two-rank tests of test_gpu_ddp.py talk over gloo).  Sums over a group of one are identities, so every
result must EQUAL the dist=None step's (the statistic accumulators are order-independent fixed point: bit for bit)."""
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
from mhentropy_amd.train import TrainStep, GraphedStep
from test_gpu_train import _model_and_state
rank, local_rank, world, dist = mdist.init("nccl")
assert dist is not None and world == 1 and dist.get_backend() == "nccl", (world, dist)
B, N = 3, 4
xn, yn = synth.batch(80, B, image_size=96)
x, y = torch.as_tensor(xn).cuda(), {k: torch.as_tensor(v).cuda() for k, v in yn.items()}
z0 = torch.as_tensor(synth.noise(80, N * B)).cuda()
rec = {}
# reference: no process group at all
m0, _ = _model_and_state("resnet18", 64, 2)
t0 = TrainStep(m0, lr=0.0)
o0 = t0.step(x, y, noise=z0, N=N)
G0, loss0 = t0.G.clone(), float(o0["total"])
# eager step over RCCL: four async bucket all-reduces issued while the reverse pass runs
m1, _ = _model_and_state("resnet18", 64, 2)
t1 = TrainStep(m1, dist=dist, lr=0.0)
assert t1.comm
t1.forward_backward(x, y, noise=z0, N=N)
rec["buckets_in_flight"] = len(t1._works)
t1.finish_allreduce()
rec["eager_grad_equal"] = bool(torch.equal(t1.G, G0))
rec["eager_grad_err"] = float((t1.G - G0).abs().max() / G0.abs().max())
o1 = t1.step(x, y, noise=z0, N=N)
rec["eager_loss"] = [float(o1["total"]), loss0]
# six HIP graphs with the collectives between them
gs = GraphedStep(t1, x, y, noise=z0, N=N)
errs = []
for _ in range(2):
    o = gs.replay(); torch.cuda.synchronize()
    errs.append(float((t1.G - G0).abs().max() / G0.abs().max()))
rec["graphs"], rec["actions"] = len(gs.graphs), [a[0] for a in gs.actions]
rec["graph_grad_err"], rec["graph_loss"] = errs, float(o["total"])
rec["graph_grad_equal"] = bool(torch.equal(t1.G, G0))
# hypothesis-sharded exchange: all_gather_into_tensor, all_reduce of the per-image sums, reduce_scatter_tensor
m2, _ = _model_and_state("resnet18", 64, 2)
t2 = TrainStep(m2, dist=dist, lr=0.0, shard_hypotheses=True)
assert t2.shard_hypotheses
o2 = t2.forward_backward(x, y, noise=z0, N=N)
t2.finish_allreduce()
rec["hyp_grad_err"] = float((t2.G - G0).abs().max() / G0.abs().max())
rec["hyp_logp_err"] = float((o2["log_p"] - o0["log_p"]).abs().max() / o0["log_p"].abs().max())
# a parameter update through the whole thing (lr > 0): finite, and the same as without the group
m3, _ = _model_and_state("resnet18", 64, 2)
m4, _ = _model_and_state("resnet18", 64, 2)
ta, tb = TrainStep(m3, lr=1e-3), TrainStep(m4, dist=dist, lr=1e-3)
ta.step(x, y, noise=z0, N=N); tb.step(x, y, noise=z0, N=N)
torch.cuda.synchronize()
rec["params_equal"] = bool(torch.equal(ta.P, tb.P))
rec["params_err"] = float((ta.P - tb.P).abs().max())
# the bf16 exchange (all_to_all_single + all_gather_into_tensor over RCCL, f32 accumulation): over one rank the gradient comes back
# rounded to bf16 once - exactly G0.bfloat16() - through the eager step and through the six-graph replay
from mhentropy_amd.dist import GradExchange
m5, _ = _model_and_state("resnet18", 64, 2)
t5 = TrainStep(m5, dist=dist, lr=0.0)
t5._xchg = GradExchange(dist, mode="bf16", device=t5.dev)
t5.forward_backward(x, y, noise=z0, N=N)
t5.finish_allreduce()
torch.cuda.synchronize()
want = G0.bfloat16().float()
rec["bf16_exchange_equal"] = bool(torch.equal(t5.G, want))
rec["bf16_exchange_err"] = float((t5.G - want).abs().max() / G0.abs().max())
gs5 = GraphedStep(t5, x, y, noise=z0, N=N)
gs5.replay(); torch.cuda.synchronize()
rec["bf16_exchange_graph_equal"] = bool(torch.equal(t5.G, want))
rec["exchange_stream_is_separate"] = bool(t5._xchg.side is not None and t5._xchg.side != torch.cuda.current_stream())
with open(os.path.join(os.environ["MHE_OUT"], "rccl.json"), "w") as fh:
    json.dump(rec, fh)
dist.barrier()
dist.destroy_process_group()
'''


def test_one_rank_rccl_group_runs_every_collective_of_the_train_step(gpu_lib, tmp_path):
    # (gpu_lib only loads the library: the child is started before this process has made a HIP call of its own that matters to it -
    # it is a separate process with its own context either way, like test_gpu_ddp.py's ranks)
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MHE_ROOT=ROOT, MHE_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", MHE_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    rec = json.load(open(tmp_path / "rccl.json"))
    assert rec["buckets_in_flight"] == 4, rec
    # a sum over one rank is the identity and every cross-workgroup sum of the step is order-independent: the gradient IS the
    # dist=None step's, bit for bit (the hypothesis-sharded step sums per-image terms in another order: a tolerance there)
    assert rec["eager_grad_equal"] and rec["hyp_grad_err"] < 2e-3 and rec["hyp_logp_err"] < 1e-5, rec
    assert rec["graphs"] == 6 and rec["actions"] == ["allreduce"] * 4 + ["wait"], rec
    assert rec["graph_grad_equal"] and max(rec["graph_grad_err"]) == 0.0, rec
    assert abs(rec["eager_loss"][0] - rec["eager_loss"][1]) <= 1e-6 * abs(rec["eager_loss"][1]), rec
    assert abs(rec["graph_loss"] - rec["eager_loss"][1]) <= 1e-6 * abs(rec["eager_loss"][1]), rec
    assert rec["params_equal"], rec
    # round 5: the bucket exchanges run on a communication stream of their own behind one event each (dist.GradExchange); the optional bf16
    # exchange returns the once-rounded gradient
    assert rec["exchange_stream_is_separate"] and rec["bf16_exchange_equal"] and rec["bf16_exchange_graph_equal"], rec
