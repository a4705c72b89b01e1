#!/usr/bin/env python3
"""Why did the run resumed from a checkpoint report twice the untrained loss?  Separates the three candidates:
(1) the --cfg-built model differs from the flag-built one, (2) TrainStep's operand packs do not follow load_model, (3) Adam restarts
(fresh moments: the first update moves EVERY parameter by lr) at a rate ten times the one the checkpoint was last trained at."""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mhentropy_amd import harness, run, synth
from mhentropy_amd.network import MHEnt
from mhentropy_amd.train import TrainStep


def batch(step, B=8, size=96):
    xn, yn = synth.batch(1000 * 0 + step, B, image_size=size)
    return torch.as_tensor(xn).cuda(), {k: torch.as_tensor(v).cuda() for k, v in yn.items()}


def main():
    tmp = tempfile.mkdtemp()
    ck = os.path.join(tmp, "ck.pth")
    log = run.main(["--backbone", "resnet18", "--batch", "8", "--hyps", "6", "--test-samples", "5", "--hidden", "64", "--flow-steps", "2",
                    "--dtype", "f32", "--epochs", "2", "--iters", "4", "--image-size", "96", "--milestones", "1", "--save", ck])
    print("run 1 per-iteration losses:", [r["it_losses"] for r in log])
    cfgp = os.path.join(tmp, "tiny.yaml")
    open(cfgp, "w").write("dataset:\n  dataset_name: ho3d\nnetwork:\n  enc_type: MHEnt\n  input: image\n  num_latent: 512\n  backbone: resnet18\n"
                          "  h_dims: [64, 64]\n  num_steps: 2\n  regressor: realnvp\n  rot_prior: null\n  w_reg_th: 50\n  w_prior_2d: 0\n  w_reg_ds: 0\n"
                          "  b_2d: 0.03\n  entropy: true\n  mode: false\ntraining:\n  batch_size: 8\n  lr: 0.0002\n  milestones: [150, 250]\n  test_samples: 5\n")

    def flag_model():
        return harness.build_mhent(backbone="resnet18", h_dims=(64, 64), num_steps=2, tables=synth.mano_tables(0)).cuda().train()

    def cfg_model():
        cfg = harness.load_config(cfgp)
        special, common = harness.mhent_cfgs_from_config(cfg, tables=synth.mano_tables(0), compute_dtype=torch.float32)
        m = MHEnt(special, **common)
        m.q_z_giv_i.compute_dtype = torch.float32
        return m.cuda().train()

    noise = torch.as_tensor(synth.noise(5, 6 * 8)).cuda()
    res = {}
    for name, mk in (("flag", flag_model), ("cfg", cfg_model)):
        torch.manual_seed(0)
        m = mk()
        harness.load_model(ck, m, map_location="cuda")
        with torch.no_grad():
            vals = []
            for step in (0, 1, 7):
                x, y = batch(step)
                vals.append(float(-m.get_loss(x, y, mods=["uv"], N=6, noise=noise)["log_p"].mean()))
        res[name] = vals
        print(f"{name}-built + load_model, no-grad get_loss on batches 0, 1, 7 (fixed noise): {vals}")
        for lr in (2e-4, 2e-5):
            torch.manual_seed(0)
            m2 = mk()
            harness.load_model(ck, m2, map_location="cuda")
            ts = TrainStep(m2, lr=lr, max_norm=1.0)
            tot = []
            for step in (0, 1, 2, 3):
                x, y = batch(step)
                tot.append(float(ts.step(x, y, noise=noise, N=6)["total"]))
            print(f"  TrainStep after load, lr {lr:g}: totals of iterations 0..3 = {[round(v, 1) for v in tot]}   (iteration 0 is the loaded model's loss)")
    print("cfg-built == flag-built after load:", np.allclose(res["flag"], res["cfg"], rtol=1e-5))
    # untrained reference point
    torch.manual_seed(0)
    m = flag_model()
    ts = TrainStep(m, lr=2e-4)
    tot = []
    for step in (0, 1, 2, 3):
        x, y = batch(step)
        tot.append(float(ts.step(x, y, noise=noise, N=6)["total"]))
    print("untrained model, lr 2e-4: totals of iterations 0..3 =", [round(v, 1) for v in tot])


if __name__ == "__main__":
    main()
