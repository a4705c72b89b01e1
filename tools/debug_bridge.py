import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from mhentropy_amd import synth
from mhentropy_amd.criteria import MHEntLoss
from mhentropy_amd.train import TrainStep
from test_gpu_train import _model_and_state
xn, yn = synth.batch(3, 4, image_size=96)
x, y = torch.as_tensor(xn).cuda(), {k: torch.as_tensor(v).cuda() for k, v in yn.items()}
z0 = torch.as_tensor(synth.noise(3, 6 * 4)).cuda()
def fused():
    m, _ = _model_and_state("resnet18", 64, 2); ts = TrainStep(m); ts.forward_backward(x, y, noise=z0, N=6); return ts.G.clone()
def bridge():
    m, _ = _model_and_state("resnet18", 64, 2); ts = TrainStep(m).attach()
    out = m.get_loss(x, y, mods=["uv"], N=6, noise=z0); total, _, _ = MHEntLoss()(dict(out), y); total.backward(); return ts.G.clone()
a, b, c, d = fused(), fused(), bridge(), bridge()
rel = lambda u, v: ((u - v).abs().max() / v.abs().max()).item()
print("fused-fused", rel(a, b), "bridge-bridge", rel(c, d), "fused-bridge", rel(a, c), rel(b, d))
