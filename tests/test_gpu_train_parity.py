"""The TIMED train path against the oracle at ResNet-50 scale (VERDICT r4 "next" #1).

`bench.py`'s `train_step` line runs the bf16 trunk + bf16 flow with every default-on path (Gram statistics and the conv3 fold of
layer1 / layer2, resident-tile 3x3, fused reverse chain, grouped weight gradients, partial slabs).  This file puts numbers on THAT
path's gradient: the flat gradient of total = mean(-log_p) (hand/criteria.py:55,173; hand/CrossModalHand.py:455-470) at 256x256,
K = 64 hypotheses per image, against torch autograd on the CPU oracle

  * in float64 - the arbiter,
  * in float32 - the reference's own arithmetic (north_star's 1e-4 bar is relative to it), and
  * in float32 with the trunk's STORED tensors rounded to bf16 (oracle/resnet_ref.forward_bf16_storage under autograd: weights, raw
    convolution outputs, activations and every gradient that passes one of those points; all arithmetic f32, the flow and the heads
    exact) - what bf16 storage alone does to the gradient, independent of any kernel.

What the figures say (tools/bf16_grad_sensitivity.py): the reference's loss is not a smooth function of the forward
values - a Laplace likelihood (gradient sign(y - mu) / b, hand/network.py:233-258) behind 49 ReLU gates and a max pool - so a forward
error eps moves the gradient by O(sqrt(eps)) per gate layer, not O(eps).  On the bench's own model (He-normal random init, the
reference loads ImageNet weights that cannot be fetched here) the f32 and f64 ORACLES already differ by 2e-2 in the trunk, and bf16
storage decorrelates the trunk gradient from the exact one (cosine 0.1-0.4) in the emulation and in the HIP step alike; with damped
residual branches (bn3 gamma x 0.15: a smoother, trained-like trunk) both reach cosine 0.94-0.99.  Hence three kinds of assertion:

  f32 mode    |HIP - f64| <= max(1e-4, 2 |f32 oracle - f64|) per parameter group (relative L2) - the product's 1e-4 path;
  bf16 mode   heads and flow (no shattering in front of them) at ~2x the measured figures; the trunk groups NO FARTHER from f64 than
              the bf16-storage oracle is (x 1.25 + 0.03), i.e. the kernels add nothing to what the dtype costs;
  both        gradient norm, loss value, and the parameters after one clip_grad_norm_(1.0) + Adam step.

Measured figures are printed and written to gpurun_out/r05_train_parity.json when that directory exists (committed copy:
profiles/r05_train_parity.json)."""
import json
import os

import pytest
import torch

from mhentropy_amd import synth

pytestmark = pytest.mark.gpu

GROUPS = [("stem", ("feat_extractor.res.conv1.", "feat_extractor.res.bn1.")), ("layer1", ("feat_extractor.res.layer1.",)),
          ("layer2", ("feat_extractor.res.layer2.",)), ("layer3", ("feat_extractor.res.layer3.",)),
          ("layer4", ("feat_extractor.res.layer4.",)), ("l1", ("feat_extractor.l1.",)), ("det_head", ("det_head.",)),
          ("flow.s", ("q_z_giv_i.s.",)), ("flow.t", ("q_z_giv_i.t.",))]
TRUNK = ("stem", "layer1", "layer2", "layer3", "layer4")


def _group_of(name):
    for g, prefixes in GROUPS:
        if name.startswith(prefixes):
            return g
    return None


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-300))


def _cos(a, b):
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


def _cat(d, names):
    return torch.cat([d[n].reshape(-1).double() for n in names])


_CASES = {}


def _case(init):
    """ResNet-50 + the shipped flow at 256x256, K = 64.  init 'bench': bench.py's own random initialisation, B = 32 (oracle in f64, f32
    and f32 with bf16 storage: ~25 + 16 + 20 s of CPU time); init 'damped': the same with bn3's gamma x 0.15, B = 16 (f64 + bf16 storage)"""
    if init in _CASES:
        return _CASES[init]
    from oracle import train_ref, mano_ref
    B, N = (32, 64) if init == "bench" else (16, 64)
    sd = {}
    sd.update({"feat_extractor.res." + k: v for k, v in synth.resnet_state(50, "resnet50").items()})
    sd.update(synth.head_state(51, 2048))
    sd.update({"q_z_giv_i." + k: v for k, v in synth.flow_state(52, 45, 512, (512, 512), 6).items()})
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    if init == "damped":
        for k in sd:
            if k.endswith("bn3.weight"):
                sd[k] = sd[k] * 0.15
    xn, yn = synth.batch(53, B, image_size=256)
    z0 = torch.as_tensor(synth.noise(53, N * B))
    x, y = torch.as_tensor(xn), {k: torch.as_tensor(v) for k, v in yn.items()}
    res = {}
    runs = [("f64", torch.float64, False), ("bf16s", torch.float32, True)] + ([("f32", torch.float32, False)] if init == "bench" else [])
    for tag, DT, emu in runs:
        tb = mano_ref.tables_from_numpy(synth.mano_tables(0), dtype=DT)
        cast = lambda d: {k: (v.to(DT) if v.is_floating_point() else v) for k, v in d.items()}
        out, total, grads, _ = train_ref.loss_and_grads(cast(sd), tb, x.to(DT), cast(y), z0.to(DT), N, arch="resnet50", bf16_storage=emu)
        res[tag] = dict(total=float(total), grads={k: v.detach() for k, v in grads.items()})
    _CASES[init] = dict(B=B, N=N, sd=sd, x=x, y=y, z0=z0, ref=res)
    return _CASES[init]


def _build(c, dtype):
    from mhentropy_amd import harness
    model = harness.build_mhent(backbone="resnet50", h_dims=(512, 512), num_steps=6, tables=synth.mano_tables(0), compute_dtype=dtype)
    missing, unexpected = model.load_state_dict(c["sd"], strict=False)
    assert not unexpected and all(k.startswith("mano_dec") for k in missing)
    return model.cuda().train()


def _report(got, c):
    """got: {name: gradient (cpu)}.  Per group and whole: relative L2 / cosine against the f64 oracle, next to the oracle's own f32 and
    bf16-storage runs"""
    ref = c["ref"]
    g64 = ref["f64"]["grads"]
    names = [n for n in got if n in g64 and _group_of(n) is not None and float(g64[n].abs().max()) > 0]
    rep = {}
    for g, _ in GROUPS + [("ALL", None)]:
        ns = names if g == "ALL" else [n for n in names if _group_of(n) == g]
        a, b64 = _cat(got, ns), _cat(g64, ns)
        r = {"tensors": len(ns), "elements": int(a.numel()), "rel_l2_vs_f64": _rel(a, b64), "cosine_vs_f64": _cos(a, b64),
             "norm_ratio": float(a.norm() / b64.norm())}
        for tag in ("f32", "bf16s"):
            if tag in ref:
                b = _cat(ref[tag]["grads"], ns)
                r[f"{tag}_oracle_rel_l2_vs_f64"], r[f"{tag}_oracle_cosine_vs_f64"] = _rel(b, b64), _cos(b, b64)
                r[f"rel_l2_vs_{tag}_oracle"] = _rel(a, b)
        rep[g] = r
    return rep


def _dump(tag, rep):
    for g, r in rep.items():
        if isinstance(r, dict) and "rel_l2_vs_f64" in r:
            extra = "".join(f"  [{t} oracle: {r[t + '_oracle_rel_l2_vs_f64']:.3e} / {r[t + '_oracle_cosine_vs_f64']:.5f}]" for t in ("f32", "bf16s")
                            if t + "_oracle_rel_l2_vs_f64" in r)
            print(f"  {tag:12s} {g:9s} rel-L2 / cosine vs f64: {r['rel_l2_vs_f64']:.3e} / {r['cosine_vs_f64']:.5f}{extra}  |g|/|g64| {r['norm_ratio']:.4f}")
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        path = os.path.join(d, "r05_train_parity.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old[tag] = rep
        json.dump(old, open(path, "w"), indent=1)


def _adam_report(ts, model, got, c):
    """one optimizer step: clip_grad_norm_(1.0) + Adam(lr 2e-4) (hand/CrossModalHand.py:457-470, 191-203) against the oracle's on its f64
    (and f32) gradients: Adam's first update is lr * sign(g) wherever |g| >> eps, so the figures are sign agreement and the largest
    parameter deviation in units of lr"""
    from oracle import train_ref
    ref = c["ref"]
    p0 = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    ts.optimizer_step()
    torch.cuda.synchronize()
    p1 = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    names = [n for n in p0 if n in ref["f64"]["grads"] and _group_of(n) is not None and float(ref["f64"]["grads"][n].abs().max()) > 0]
    upd = {}
    for tag in [t for t in ("f32", "f64") if t in ref]:
        DT = torch.float32 if tag == "f32" else torch.float64
        new, norm = train_ref.clip_and_adam({n: p0[n].to(DT) for n in names}, {n: ref[tag]["grads"][n] for n in names}, {})
        upd[tag] = ({n: (new[n] - p0[n].to(DT)).double() for n in names}, float(norm))
    d_hip = {n: (p1[n].double() - p0[n].double()) for n in names}
    a, b64 = _cat(d_hip, names), _cat(upd["f64"][0], names)
    r = {"update_rel_l2_vs_f64": _rel(a, b64), "update_sign_mismatch_frac": float(((a * b64) < 0).double().mean()),
         "max_param_deviation_over_lr": float((a - b64).abs().max() / 2e-4),
         "grad_norm_hip": float(torch.sqrt(sum((got[n].double() ** 2).sum() for n in got))), "grad_norm_f64": upd["f64"][1]}
    if "f32" in upd:
        b32 = _cat(upd["f32"][0], names)
        r.update({"f32_oracle_update_rel_l2_vs_f64": _rel(b32, b64), "f32_oracle_sign_mismatch_frac": float(((b32 * b64) < 0).double().mean())})
    return r


def _run(c, mode):
    from mhentropy_amd.train import TrainStep
    model = _build(c, torch.bfloat16 if mode == "bf16" else torch.float32)
    ts = TrainStep(model)
    if mode == "bf16":       # the timed configuration: every default-on path of the bf16 step really is on
        assert ts.flow_bf16 and ts.train_recompute and ts.conv3_fold and ts.conv_halo and ts.conv_halo_dg and ts.gate_bits
    xg, yg = c["x"].cuda(), {k: v.cuda() for k, v in c["y"].items()}
    out = ts.forward_backward(xg, yg, noise=c["z0"].cuda(), N=c["N"])
    assert torch.isfinite(ts.G).all()
    got = {n: ts.grad_of(p).detach().cpu().clone() for n, p in model.named_parameters()}
    rep = _report(got, c)
    t64 = c["ref"]["f64"]["total"]
    rep["loss_rel_vs_f64"] = abs(float(out["total"]) - t64) / abs(t64)
    for tag in ("f32", "bf16s"):
        if tag in c["ref"]:
            rep[f"{tag}_oracle_loss_rel_vs_f64"] = abs(c["ref"][tag]["total"] - t64) / abs(t64)
    rep["adam"] = _adam_report(ts, model, got, c)
    return rep


def test_f32_train_step_gradient_at_resnet50_scale_within_the_f32_oracles_own_distance_from_f64(gpu_lib):
    c = _case("bench")
    rep = _run(c, "f32")
    print(f"\nTrainStep (f32) vs oracle, ResNet-50 256x256, B={c['B']}, K={c['N']}: loss rel {rep['loss_rel_vs_f64']:.3e} "
          f"(f32 oracle: {rep['f32_oracle_loss_rel_vs_f64']:.3e})")
    _dump("f32/bench", rep)
    print("  adam:", {k: f"{v:.4g}" for k, v in rep["adam"].items()})
    for g, _ in GROUPS + [("ALL", None)]:
        r = rep[g]
        assert r["rel_l2_vs_f64"] <= max(1e-4, 2.0 * r["f32_oracle_rel_l2_vs_f64"]), (g, r)
    assert rep["loss_rel_vs_f64"] <= max(1e-4, 2.0 * rep["f32_oracle_loss_rel_vs_f64"])
    a = rep["adam"]
    assert abs(a["grad_norm_hip"] - a["grad_norm_f64"]) <= 1e-3 * a["grad_norm_f64"]
    assert a["update_sign_mismatch_frac"] <= max(1e-3, 2.0 * a["f32_oracle_sign_mismatch_frac"]), a


# heads and flow of the bf16 step, relative L2 against f64: measured on MI355X (profiles/r05_train_parity.json) l1 4.1e-2, det_head 1.2e-2,
# flow.s 3.4e-2, flow.t 6.0e-2 on the bench's initialisation; bounds ~2x
BF16_HEAD_BOUND = {"l1": 0.08, "det_head": 0.03, "flow.s": 0.07, "flow.t": 0.12}


@pytest.mark.parametrize("init", ["bench", "damped"])
def test_bf16_train_step_gradient_at_resnet50_scale_against_the_f64_and_bf16_storage_oracles(gpu_lib, init):
    c = _case(init)
    rep = _run(c, "bf16")
    print(f"\nTrainStep (bf16, {init} initialisation) vs oracle, ResNet-50 256x256, B={c['B']}, K={c['N']}: loss rel {rep['loss_rel_vs_f64']:.3e} "
          f"(bf16-storage oracle: {rep['bf16s_oracle_loss_rel_vs_f64']:.3e})")
    _dump("bf16/" + init, rep)
    print("  adam:", {k: f"{v:.4g}" for k, v in rep["adam"].items()})
    for g, bound in BF16_HEAD_BOUND.items():
        assert rep[g]["rel_l2_vs_f64"] < bound, (g, rep[g])
    for g in TRUNK + ("ALL",):
        r = rep[g]
        # the kernels add nothing to what bf16 storage costs: no farther from f64 than the emulation, up to its own run-to-run scale
        assert r["rel_l2_vs_f64"] <= 1.25 * r["bf16s_oracle_rel_l2_vs_f64"] + 0.03, (g, r)
        assert r["cosine_vs_f64"] >= r["bf16s_oracle_cosine_vs_f64"] - 0.08, (g, r)
        assert 0.9 < r["norm_ratio"] < 1.1, (g, r)
    if init == "damped":      # the well-conditioned case: the direction holds
        assert rep["ALL"]["cosine_vs_f64"] > 0.9 and rep["layer4"]["cosine_vs_f64"] > 0.97, rep["ALL"]
    assert rep["loss_rel_vs_f64"] < 1e-3
    a = rep["adam"]
    assert abs(a["grad_norm_hip"] - a["grad_norm_f64"]) <= 3e-2 * a["grad_norm_f64"], a
