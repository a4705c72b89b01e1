import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from mhentropy_amd import resnet, synth, ops
from oracle import resnet_ref
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
B, S = 8, 128
sdn = synth.resnet_state(4, arch)
sd = {k: torch.as_tensor(v) for k, v in sdn.items()}
x, _ = synth.batch(4, B, image_size=S)
trunk = resnet.ResNetTrunk(arch, compute_dtype=torch.bfloat16)
trunk.load_state_dict(sd)
trunk = trunk.cuda().train()
taps = []
orig = ops.bn_act
def hook(x_, scale, shift, res=None, res_scale=None, res_shift=None, relu=True, out=None):
    y = orig(x_, scale, shift, res, res_scale, res_shift, relu, out)
    if res is not None:
        taps.append(y.float().cpu().permute(0, 3, 1, 2).clone())
    return y
ops.bn_act = hook
f = trunk(torch.as_tensor(x).cuda())
# oracle with taps
q = resnet_ref._q
kind, blocks, _ = resnet_ref.CFG[arch]
ref_taps = []
def conv_bn(inp, cname, bname, stride=1, pad=0):
    y = F.conv2d(inp, q(sd[cname + ".weight"]), None, stride, pad)
    mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    sc = sd[bname + ".weight"] / torch.sqrt(var + 1e-5); sh = sd[bname + ".bias"] - mean * sc
    return q(y), sc.view(1, -1, 1, 1), sh.view(1, -1, 1, 1)
y, sc, sh = conv_bn(q(torch.as_tensor(x)), "conv1", "bn1", 2, 3)
a = q(F.max_pool2d(F.relu(y * sc + sh), 3, 2, 1))
i = 0
for li, nb in enumerate(blocks):
    for bi in range(nb):
        stride = 2 if (bi == 0 and li > 0) else 1
        p = f"layer{li + 1}.{bi}"
        a_in = a
        y, sc, sh = conv_bn(a, p + ".conv1", p + ".bn1")
        y, sc, sh = conv_bn(q(F.relu(y * sc + sh)), p + ".conv2", p + ".bn2", stride, 1)
        y, sc, sh = conv_bn(q(F.relu(y * sc + sh)), p + ".conv3", p + ".bn3")
        if (p + ".downsample.0.weight") in sd:
            yd, scd, shd = conv_bn(a, p + ".downsample.0", p + ".downsample.1", stride)
            idt = yd * scd + shd
        else:
            idt = a
        a = q(F.relu(y * sc + sh + idt))
        t = taps[i]; i += 1
        e = (t - a).abs()
        # feed-forward check: product block applied to the ORACLE's input would isolate per-block error; here cumulative
        print(f"{p:12s} scale {a.abs().max():9.3f} mean|a| {a.abs().mean():8.4f} max-err {e.max():9.4f} mean-err {e.mean():9.5f} rel-mean {e.mean()/a.abs().mean():.3e}")
