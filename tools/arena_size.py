import sys, torch
sys.path.insert(0, "/root/repo")
from mhentropy_amd import harness, synth
from mhentropy_amd.train import TrainStep
model = harness.build_mhent(backbone="resnet50", tables=synth.mano_tables(0), compute_dtype=torch.bfloat16).cuda().train()
ts = TrainStep(model)
for dt, a in ts._arena.items():
    print(dt, a["used"] / 1e6, "M elements; params", ts.n_params / 1e6)
