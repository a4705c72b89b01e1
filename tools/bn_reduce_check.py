import sys, torch
sys.path.insert(0, "/root/repo")
from mhentropy_amd import ops
torch.manual_seed(0)
for (P, C) in ((16384, 2048), (65536, 256), (1000, 64)):
    g = torch.randn(P, C, device="cuda").bfloat16(); y = torch.randn(P, C, device="cuda").bfloat16()
    mi = torch.stack([torch.randn(C) * 0.1, torch.rand(C) + 0.5]).cuda().contiguous()
    gamma = (torch.rand(C) + 0.5).cuda()
    st = ops.stat_unit(C, "cuda"); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    ops.bn_backward(g.view(1, 1, P, C), None, y.view(1, 1, P, C), mi, gamma, st, dg, db, coef_only=True)
    gf, yf = g.double(), y.double()
    ref_db = gf.sum(0); ref_dg = (gf * (yf - mi[0].double()) * mi[1].double()).sum(0)
    print(P, C, "dbeta rel", ((db.double() - ref_db).abs().max() / ref_db.abs().max()).item(), "dgamma rel", ((dg.double() - ref_dg).abs().max() / ref_dg.abs().max()).item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.bn_backward(g.view(1, 1, P, C), None, y.view(1, 1, P, C), mi, gamma, st, dg, db, coef_only=True)
    e1.record(); torch.cuda.synchronize()
    print("   reduce + finalize", e0.elapsed_time(e1) * 100, "us")
