"""Train-step part for the conditional Glow branch (`q_z_giv_i_model == 'glow'`): the sampling pass
`sample_and_log_prob` with a tape, and its hand-written reverse pass - what autograd does for the reference in
`loss_ent = log_prob.mean()` (reference README.md:36-42) / hand/network.py:736-742,781-799 + CrossModalHand.py:455-470.

PARITY UNPINNED like the forward (mhentropy_amd/glow.py): checked against torch autograd on the nflows restatement
(oracle/glow_ref.py), not against the absent third-party class.

Per layer (sampling order L-1 .. 0, reverse pass 0 .. L-1):
    params = ResidualNet(v[identity columns], context);  y = (v - shift) / scale on the transform columns;  v' = Ainv y + cinv
with (A, c) the ActNorm + LU affine map.  Dense products: mhe_linear_f32 / mhe_conv_wgrad_nhwc; elementwise stages: csrc/glow.hip.
The 45x45 re-parameterisation - A, A^-1, the log-det constant from (log_scale, shift, LU entries, softplus diagonal, bias), and their
gradients from dAinv, dcinv and the constant - runs in float64 ON THE DEVICE since round 5 (csrc/glow_affine.hip: one workgroup per layer;
rounds 2-4 did it in numpy on the host, a queue drain per step that kept this branch out of HIP graphs).  Dropout (train mode, p = 0.2: hand/network.py:343-344,781) is applied to the second activation
of every residual block in the sampling pass (mask bits kept on the tape) and to its gradient in the reverse pass (glow.py, mhe_dropout).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from . import ops, _lib


class GlowPart:
    def __init__(self, ts, glow):
        self.ts, self.g = ts, glow
        D, H, Fc, L, NB = glow.features, glow.hidden, glow.context_features, glow.num_layers, glow.num_blocks
        if D > 64:
            raise NotImplementedError("the Glow reverse pass is built for the hand flow (features <= 64); the 144-D body geometry runs "
                                      "forward / sample / log_prob only")
        self.per = 1 + NB
        self.mixed = glow.compute_dtype == torch.bfloat16 and H % 64 == 0
        T = glow._transform._transforms
        slots = L * self.per
        self.raw_wctx, self.raw_bctx = ts._raw_slot((slots * H, Fc)), ts._raw_slot((slots * H,))
        self.layers = []
        wctx_idx, bctx_idx = [], []
        ar = lambda n: torch.arange(n, dtype=torch.int64)
        # the residual blocks' gradients in (layer, block) order, one pitch apart: what the grouped weight-gradient launches and the single
        # bias column sum of the fused reverse pass write ([L NB][H][H] x 2; [L NB][b0 | b1][H])
        self.raw_w0, self.raw_w1, self.raw_bias = ts._raw_slot((L * NB * H * H,)), ts._raw_slot((L * NB * H * H,)), ts._raw_slot((L * NB * 2 * H,))
        # ... and the initial / final layers' ([L][H][64], [L][64][H], [L][64]): the one-launch reverse chain's grouped launches write [L]-strided
        self.raw_wx, self.raw_wf, self.raw_bf = ts._raw_slot((L * H * 64,)), ts._raw_slot((L * 64 * H,)), ts._raw_slot((L * 64,))
        for l in range(L):
            an, lu, cp = T[3 * l], T[3 * l + 1], T[3 * l + 2]
            net = cp.transform_net
            idf = cp.identity_features.cpu()
            nid, nt = idf.numel(), int(cp.transform_features.numel())
            d = {"an": an, "lu": lu, "cp": cp, "nid": nid, "nt": nt, "idf": idf}
            # small re-parameterisation gradients land in exact-size raw slots (written from the host chain)
            for name, p in (("log_scale", an.log_scale), ("shift", an.shift), ("lower", lu.lower_entries), ("upper", lu.upper_entries),
                            ("udiag", lu.unconstrained_upper_diag), ("bias", lu.bias)):
                d["r_" + name] = ts._raw_slot(p.shape)
                ts._map_grad(p, ar(p.numel()).view(p.shape) + d["r_" + name])
            d["r_ainv"], d["r_cinv"] = ts._raw_slot((64, 64)), ts._raw_slot((64,))
            d["r_wx"], d["r_wf"], d["r_bf"] = self.raw_wx + l * H * 64, self.raw_wf + l * 64 * H, self.raw_bf + l * 64
            s0 = l * self.per
            wi = torch.empty(H, nid + Fc, dtype=torch.int64)
            wi[:, :nid] = (ar(H * 64).view(H, 64) + d["r_wx"])[:, idf]
            wi[:, nid:] = ar(H * Fc).view(H, Fc) + self.raw_wctx + s0 * H * Fc
            ts._map_grad(net.initial_layer.weight, wi)
            ts._map_grad(net.initial_layer.bias, ar(H) + self.raw_bctx + s0 * H)
            ts._map_grad(net.final_layer.weight, (ar(64 * H).view(64, H) + d["r_wf"])[:2 * nt])
            ts._map_grad(net.final_layer.bias, ar(2 * nt) + d["r_bf"])
            d["r_blocks"] = []
            for b, blk in enumerate(net.blocks):
                kb = l * NB + b
                rb = {"w0": self.raw_w0 + kb * H * H, "w1": self.raw_w1 + kb * H * H, "b0": self.raw_bias + (2 * kb) * H, "b1": self.raw_bias + (2 * kb + 1) * H}
                for j in range(2):
                    ts._map_grad(blk.linear_layers[j].weight, ar(H * H).view(H, H) + rb[f"w{j}"])
                    ts._map_grad(blk.linear_layers[j].bias, ar(H) + rb[f"b{j}"])
                ts._map_grad(blk.context_layer.weight, ar(H * Fc).view(H, Fc) + self.raw_wctx + (s0 + 1 + b) * H * Fc)
                ts._map_grad(blk.context_layer.bias, ar(H) + self.raw_bctx + (s0 + 1 + b) * H)
                d["r_blocks"].append(rb)
                # operand layouts refreshed on the device by the trainer's gather tables (like every other derived weight)
            wxi = torch.full((H, 64), -1, dtype=torch.int64)
            wxi[:, idf] = ts._pidx(net.initial_layer.weight)[:, :nid]
            wfi = torch.full((64, H), -1, dtype=torch.int64)
            wfi[:2 * nt] = ts._pidx(net.final_layer.weight)
            bfi = torch.full((64,), -1, dtype=torch.int64)
            bfi[:2 * nt] = ts._pidx(net.final_layer.bias)
            f32 = torch.float32
            d["wxi"], d["wfi"], d["bfi"] = wxi, wfi, bfi
            d["wx"], d["wxT"] = ts._derived(wxi, f32), ts._derived(wxi.t().contiguous(), f32)
            d["wf"], d["wfT"], d["bf"] = ts._derived(wfi, f32), ts._derived(wfi.t().contiguous(), f32), ts._derived(bfi, f32)
            d["blocks"] = [(blk.linear_layers[0].weight.data, blk.linear_layers[0].bias.data, blk.linear_layers[1].weight.data,
                            blk.linear_layers[1].bias.data) for blk in net.blocks]
            d["blocksT"] = [(ts._derived(ts._pidx(blk.linear_layers[0].weight).t().contiguous(), f32),
                             ts._derived(ts._pidx(blk.linear_layers[1].weight).t().contiguous(), f32)) for blk in net.blocks]
            d["first"], d["T"] = int(cp.transform_features[0]), nt
            if self.mixed:      # operands of the hidden x hidden products on bf16 MFMA (performance mode)
                bf = torch.bfloat16
                d["blocks_b"] = [(ts._derived(ts._pidx(blk.linear_layers[0].weight), bf), ts._derived(ts._pidx(blk.linear_layers[1].weight), bf),
                                  ts._derived(ts._pidx(blk.linear_layers[0].weight).t().contiguous(), bf),
                                  ts._derived(ts._pidx(blk.linear_layers[1].weight).t().contiguous(), bf)) for blk in net.blocks]
            wctx_idx.append(ts._pidx(net.initial_layer.weight)[:, nid:]); bctx_idx.append(ts._pidx(net.initial_layer.bias))
            for blk in net.blocks:
                wctx_idx.append(ts._pidx(blk.context_layer.weight)); bctx_idx.append(ts._pidx(blk.context_layer.bias))
            self.layers.append(d)
        self.wctx = ts._derived(torch.cat(wctx_idx), torch.float32)
        self.bctx = ts._derived(torch.cat(bctx_idx), torch.float32)
        self.wctxT = ts._derived(torch.cat(wctx_idx).t().contiguous(), torch.float32)
        # the six small tensors of every layer (in the flat parameter buffer) -> A, A^-1, c, c^-1, constant parts: one launch per refresh
        self._ptab = torch.tensor([[p.data_ptr() for p in (d["an"].log_scale, d["an"].shift, d["lu"].lower_entries, d["lu"].upper_entries,
                                                           d["lu"].unconstrained_upper_diag, d["lu"].bias)] for d in self.layers],
                                  dtype=torch.int64, device=ts.dev)
        self.aff = ops.glow_affine(self._ptab, L, D, self.layers[0]["lu"].eps)
        self._gtabs = None
        self._tp = None
        # operands of the one-launch sampling kernel (csrc/glow_fwd.hip): gather tables like every other derived layout
        self.fused = None
        if self.mixed and H == 512 and NB == 2 and D <= 48:
            nets = [d["cp"].transform_net for d in self.layers]
            st = lambda f: torch.stack([f(n) for n in nets])
            pi = ts._pidx
            fp = ops.glow_fused_layout(torch.stack([d["wxi"] for d in self.layers]),
                                       st(lambda n: torch.stack([pi(b.linear_layers[0].weight) for b in n.blocks])),
                                       st(lambda n: torch.stack([pi(b.linear_layers[1].weight) for b in n.blocks])),
                                       torch.stack([d["wfi"] for d in self.layers]), torch.stack([d["bfi"] for d in self.layers]),
                                       st(lambda n: torch.stack([pi(b.linear_layers[0].bias) for b in n.blocks])),
                                       st(lambda n: torch.stack([pi(b.linear_layers[1].bias) for b in n.blocks])), D)
            self.fused = {k: ts._derived(v.contiguous(), torch.bfloat16 if (k.endswith("F") or k.endswith("T")) else torch.float32) for k, v in fp.items()}
            # the final layer's rows move to the flow variable's columns in the chain's [g_shift | g_us] gradient: row c of the first / second half
            # is the shift / scale row of transform column c -> back to nflows' [shift (T) | scale (T)] rows by one index_select per step
            perm = torch.zeros(L, 64, dtype=torch.int64)
            for l in range(L):
                first = 1 - (l & 1)
                T_ = D // 2 if first else (D + 1) // 2
                cols = torch.arange(first, D, 2)
                perm[l, :T_], perm[l, T_:2 * T_] = cols, 64 + cols
                perm[l, 2 * T_:] = 63                          # (a zero row of the first half: columns >= dim are never transform columns)
            self._wf_perm = (perm + 128 * torch.arange(L)[:, None]).reshape(-1).to(ts.dev)
        glow._external_pack = self.module_pack

    # ------------------------------------------------------------------ the 45x45 affine maps (float64 on the device, tiny)
    def refresh_affine(self):
        """A = W diag(exp(log_scale)), c = W shift + b, their inverse (padded to 64) and the per-layer log-det constants from the CURRENT
        parameters: mhe_glow_affine_f64 into the same tensors (graph-capturable: no host value depends on the parameters)"""
        ops.glow_affine(self._ptab, self.g.num_layers, self.g.features, self.layers[0]["lu"].eps, out=self.aff)

    def invalidate(self):
        pass             # (nothing host-side follows the parameters any more; the operand layouts are refreshed by the trainer's gather)

    def module_pack(self):
        """the operand dict ConditionalGlow._run reads (glow.py:_packed), on the trainer's device-resident layouts: the modules' own forward /
        sample / log_prob paths follow the optimizer without any host-side re-packing"""
        self.ts.sync()
        self.refresh_affine()
        pk = {"layers": [], "const_parts": self.aff["const_parts"], "aff": self.aff, "wctx": self.wctx, "bctx": self.bctx, "fused": self.fused}
        for l, d in enumerate(self.layers):
            e = {"A": self.aff["A"][l], "c": self.aff["c"][l], "Ainv": self.aff["Ainv"][l], "cinv": self.aff["cinv"][l], "wx": d["wx"],
                 "blocks": d["blocks"], "wf": d["wf"], "bf": d["bf"], "T": d["T"], "first": d["first"]}
            if self.mixed:
                e["blocks_bf16"] = [bb[:2] for bb in d["blocks_b"]]
            else:
                e["blocks_bf16"] = [(w0.to(torch.bfloat16), w1.to(torch.bfloat16)) for (w0, _, w1, _) in d["blocks"]] if self.g.compute_dtype == torch.bfloat16 else None
            pk["layers"].append(e)
        return pk

    # ------------------------------------------------------------------ sampling pass with tape
    def _forward_fused(self, z0, feat):
        """the sampling pass with its tape from ONE launch (mhe_glow_layers_bf16): the reverse pass below reads the kernel's tape tensors"""
        ts, g = self.ts, self.g
        D, H, B, R, L, NB = g.features, g.hidden, feat.shape[0], z0.shape[0], g.num_layers, g.num_blocks
        self.refresh_affine()
        ctab = ops.linear(feat, self.wctx, self.bctx)
        bits = g._drop_bits(R)
        bf = torch.bfloat16
        tape = {"v": ts._buf("glow_v", (L, R, 64)), "y": ts._buf("glow_y", (L, R, 64)), "prm": ts._buf("glow_prm", (L, R, 64)),
                "tb": ts._buf("glow_tb", (L, NB, R, H), bf), "t2": ts._buf("glow_t2", (L, NB, R, H), bf), "t3": ts._buf("glow_t3", (L, NB, R, H), bf),
                "hf": ts._buf("glow_hf", (L, R, H), bf)}
        import os
        chain = (os.environ.get("MHE_GLOW_REV_FUSED", "1") == "1" and ops.glow_reverse_chain_supported(R, B, D, H, L, NB))
        if chain:           # what the one-launch reverse chain reads besides t3: parameters in column order, the bf16 layer input, the ReLU gates as bits
            tape.update({"prmc": ts._buf("glow_prmc", (L, R, 128)), "vb": ts._buf("glow_vb", (L, R, 64), bf),
                         "bits": ts._buf("glow_bits", (L, NB, 2, B, 512, 2), torch.int32)})
        x, logq = ops.glow_layers(z0, ctab, self.fused, self.aff, bits, g.p_drop, R // B, B, D, B, 1, tape=tape)
        self._tp = {"fused": tape, "bits": bits, "ctab": ctab, "feat": feat, "chain": chain}
        return x, logq

    def forward(self, z0, feat):
        ts, g = self.ts, self.g
        L_, D, H, B, R = _lib.lib(), g.features, g.hidden, feat.shape[0], z0.shape[0]
        import os
        if (self.fused is not None and os.environ.get("MHE_GLOW_FUSED", "1") == "1" and R % B == 0
                and ops.glow_layers_supported(R // B, B, D, H, g.num_layers, g.num_blocks)):
            return self._forward_fused(z0, feat)
        self.refresh_affine()
        s, dev = ops._stream, z0.device
        ctab = ops.linear(feat, self.wctx, self.bctx)
        cs = ctab.shape[1]
        v = torch.empty(R, 64, device=dev)
        ops.check(L_.mhe_pad64_f32(ops._ptr(z0), ops._ptr(v), R, D, s()), "mhe_pad64_f32")
        zp = v
        logdet = torch.zeros(R, device=dev)
        tape = [None] * g.num_layers
        for l in range(g.num_layers - 1, -1, -1):
            d = self.layers[l]
            slot = l * self.per
            h = ops.linear(v, d["wx"])
            ops.check(L_.mhe_glow_add_image_rows_f32(ops._ptr(h), C.c_void_p(ctab[:, slot * H:].data_ptr()), cs, R, H, 1, B, s()), "mhe_glow_add_image_rows_f32")
            hs, t2s, t3s, drops = [h], [], [], []
            for b, (w0, b0, w1, b1) in enumerate(d["blocks"]):
                if self.mixed:
                    w0b, w1b = d["blocks_b"][b][:2]
                    t = torch.empty(R, 1, 1, H, device=dev, dtype=torch.bfloat16)
                    ops.check(L_.mhe_relu_copy_f32(ops._ptr(hs[-1]), ops._ptr(t), t.numel(), ops.BF16, s()), "mhe_relu_copy_f32")
                    t2 = ops.conv2d_nhwc(t, w0b, 1, 1, 1, 0, out_shift=b0, relu_out=True)
                    drops.append(g.dropout_(t2))
                    t3 = ops.conv2d_nhwc(t2, w1b, 1, 1, 1, 0, out_shift=b1)
                else:
                    t = torch.empty_like(h)
                    ops.check(L_.mhe_relu_copy_f32(ops._ptr(hs[-1]), ops._ptr(t), t.numel(), 0, s()), "mhe_relu_copy_f32")
                    t2 = ops.linear(t, w0, b0, relu=True)
                    drops.append(g.dropout_(t2))
                    t3 = ops.linear(t2, w1, b1)
                hn = hs[-1].clone()
                ops.check(L_.mhe_glow_glu_residual_f32(ops._ptr(hn), ops._ptr(t3), ops.dtype_code(t3.dtype), C.c_void_p(ctab[:, (slot + 1 + b) * H:].data_ptr()), cs, R, H, 1, B, s()),
                          "mhe_glow_glu_residual_f32")
                hs.append(hn); t2s.append(t2); t3s.append(t3)
            prm = ops.linear(hs[-1], d["wf"], d["bf"])
            y = torch.empty(R, 64, device=dev)
            ops.check(L_.mhe_glow_coupling_f32(ops._ptr(v), ops._ptr(prm), ops._ptr(y), ops._ptr(logdet), R, D, d["first"], d["T"], 1, s()),
                      "mhe_glow_coupling_f32")
            tape[l] = {"v": v, "hs": hs, "t2": t2s, "t3": t3s, "prm": prm, "y": y, "drop": drops}
            v = ops.linear(y, self.aff["Ainv"][l], self.aff["cinv"][l])
        x, logq = ops.glow_finish(zp, v, logdet, R, D, True, self.aff["const_parts"])
        self._tp = {"tape": tape, "ctab": ctab, "feat": feat}
        return x, logq

    # ------------------------------------------------------------------ reverse pass
    def _backward_chain(self, g_x, g_logp, N, B):
        """the reverse pass with the data-gradient chain of all layers in ONE launch (mhe_glow_reverse_chain_bf16, csrc/glow_rev.hip): what is
        left around it are the weight gradients - grouped launches over the tape and the chain's outputs as they lie ([L]- / [L, 2]-strided) -,
        the 45 x 45 products for dA^-1, the column sums of the per-image rows and the float64 re-parameterisation kernel"""
        ts, g = self.ts, self.g
        D, H, R, L, NB = g.features, g.hidden, g_x.shape[0], g.num_layers, g.num_blocks
        tp = self._tp
        ft, ctab = tp["fused"], tp["ctab"]
        raw, cs, bf = ts._raw, ctab.shape[1], torch.bfloat16
        out = {"gv": ts._buf("glow_gv", (L, R, 64)), "gpc": ts._buf("glow_gpc", (L, R, 128), bf), "gt3": ts._buf("glow_gt3", (L, NB, R, H), bf),
               "gt2": ts._buf("glow_gt2", (L, NB, R, H), bf), "gh0": ts._buf("glow_gh0", (L, R, H), bf), "gct": ts._buf("glow_Gct", (B, cs)),
               "bsum": ts._buf("glow_bsum", (B, L * NB * 2 * H)), "bfsum": ts._buf("glow_bfsum", (B, L * 128))}
        ops.glow_reverse_chain(g_x, g_logp, -1.0 / N, ft, ctab, self.fused, self.aff, g.p_drop if tp["bits"] is not None else 0.0, out, B, D)
        for l in range(L):          # dA^-1 = gv^T y, dc^-1 = sum gv (f32: they feed the float64 re-parameterisation)
            rs = self.layers[l]
            ops.linear_wgrad(ft["y"][l], out["gv"][l], raw(rs["r_ainv"], (64, 64))); ops.colsum(out["gv"][l], raw(rs["r_cinv"], (64,)))
        ops.conv_wgrad_batched(ft["t2"].view(L * NB, R, H), out["gt3"].view(L * NB, R, H), raw(self.raw_w1, (H, H)), H * H, L * NB)
        ops.conv_wgrad_batched(ft["tb"].view(L * NB, R, H), out["gt2"].view(L * NB, R, H), raw(self.raw_w0, (H, H)), H * H, L * NB)
        ops.conv_wgrad_batched(ft["vb"], out["gh0"], raw(self.raw_wx, (H, 64)), H * 64, L)                      # dWx[l] = gh0^T v   [H, 64]
        wfp = ts._buf("glow_wfp", (L * 128, H)); wfp.zero_()
        ops.conv_wgrad_batched(ft["hf"], out["gpc"], wfp[:128], 128 * H, L)                                     # [g_shift | g_us]^T h  [128, H] per layer
        torch.index_select(wfp, 0, self._wf_perm, out=raw(self.raw_wf, (L * 64, H)))                            # -> nflows' [shift | scale] rows
        bfp = ts._buf("glow_bfp", (L * 128,)); bfp.zero_()
        ops.colsum(out["bfsum"], bfp)
        torch.index_select(bfp, 0, self._wf_perm, out=raw(self.raw_bf, (L * 64,)))
        ops.colsum(out["bsum"], raw(self.raw_bias, (L * NB * 2 * H,)))
        Gct = out["gct"]
        ops.linear_wgrad(tp["feat"], Gct, raw(self.raw_wctx, (cs, g.context_features))); ops.colsum(Gct, raw(self.raw_bctx, (cs,)))
        g_feat = ops.linear(Gct, self.wctxT)
        self._reparam_backward(g_logp)
        return g_feat

    def _backward_fused(self, g_x, g_logp, N, B):
        """the reverse pass over the one-launch kernel's tape: per layer the small stages as before, per residual block two products on bf16
        MFMA and three per-image kernels (gate / dropout + ReLU reverse with the per-image sums inside, csrc/glow.hip); the 16 hidden x hidden
        weight gradients as TWO grouped launches after the chain (x = the tape's [L, 2, R, 512] tensors as they lie), all 16 bias gradients
        as ONE column sum of the per-image rows"""
        ts, g = self.ts, self.g
        L_, D, H, R, L, NB = _lib.lib(), g.features, g.hidden, g_x.shape[0], g.num_layers, g.num_blocks
        tp = self._tp
        ft, bits, ctab = tp["fused"], tp["bits"], tp["ctab"]
        s, dev, raw, cs, bf = ops._stream, g_x.device, ts._raw, ctab.shape[1], torch.bfloat16
        gv = torch.empty(R, 64, device=dev)
        ops.check(L_.mhe_pad64_f32(ops._ptr(g_x), ops._ptr(gv), R, D, s()), "mhe_pad64_f32")
        Gct = ts._buf("glow_Gct", (B, cs)); Gct.zero_()
        gt3_all, gt2_all = ts._buf("glow_gt3", (L, NB, R, H), bf), ts._buf("glow_gt2", (L, NB, R, H), bf)
        bs_w = L * NB * 2 * H
        bsum = ts._buf("glow_bsum", (B, bs_w))                  # per-image rows of all 16 bias gradients: every slice is written below
        dscale = 1.0 / (1.0 - g.p_drop) if bits is not None else 1.0
        for l in range(L):
            d = rs = self.layers[l]
            slot = l * self.per
            y, v, prm, hf = ft["y"][l], ft["v"][l], ft["prm"][l], ft["hf"][l]
            ops.linear_wgrad(y, gv, raw(rs["r_ainv"], (64, 64))); ops.colsum(gv, raw(rs["r_cinv"], (64,)))
            gy = ops.linear(gv, self.aff["AinvT"][l])
            gvc, gprm = torch.empty(R, 64, device=dev), torch.empty(R, 64, device=dev)
            ops.check(L_.mhe_glow_coupling_inv_bwd_f32(ops._ptr(v), ops._ptr(prm), ops._ptr(gy), ops._ptr(g_logp), -1.0 / N,
                                                       ops._ptr(gvc), ops._ptr(gprm), R, B, D, d["first"], d["T"], s()), "mhe_glow_coupling_inv_bwd_f32")
            # the final layer's operand was kept as bf16: its weight gradient on bf16 operands, f32 accumulation
            ops.conv_wgrad(hf.view(R, 1, 1, H), gprm.to(bf).view(R, 1, 1, 64), 1, 1, 1, 0, raw(rs["r_wf"], (64, H)))
            ops.colsum(gprm, raw(rs["r_bf"], (64,)))
            gh = ops.linear(gprm, d["wfT"])
            for b in range(NB - 1, -1, -1):
                kb = l * NB + b
                _, _, w0Tb, w1Tb = d["blocks_b"][b]
                gt3, gt2 = gt3_all[l, b].view(R, 1, 1, H), gt2_all[l, b].view(R, 1, 1, H)
                gate = C.c_void_p(ctab[:, (slot + 1 + b) * H:].data_ptr())
                ops.check(L_.mhe_glow_glu_bwd_sum(ops._ptr(gh), ops._ptr(ft["t3"][l, b]), gate, cs, ops._ptr(gt3),
                                                  C.c_void_p(Gct[:, (slot + 1 + b) * H:].data_ptr()), cs,
                                                  C.c_void_p(bsum[:, (2 * kb + 1) * H:].data_ptr()), bs_w, N, B, H, s()), "mhe_glow_glu_bwd_sum")
                ops.conv2d_nhwc(gt3, w1Tb, 1, 1, 1, 0, out=gt2)
                # dropout's and the ReLU's reverse in one pass: t2 = dropout(relu(.)) is zero exactly where either gate is closed
                ops.check(L_.mhe_glow_mask_scale_sum(ops._ptr(gt2), ops._ptr(ft["t2"][l, b]), dscale, C.c_void_p(bsum[:, (2 * kb) * H:].data_ptr()),
                                                     bs_w, N, B, H, s()), "mhe_glow_mask_scale_sum")
                gt = ops.conv2d_nhwc(gt2, w0Tb, 1, 1, 1, 0)
                ops.check(L_.mhe_relu_bwd_add_mixed(ops._ptr(gh), ops._ptr(gt), ops._ptr(ft["tb"][l, b]), gh.numel(), ops.BF16, ops.BF16, s()),
                          "mhe_relu_bwd_add_mixed")
            ops.linear_wgrad(v, gh, raw(rs["r_wx"], (H, 64)))
            ops.sum_over_hypotheses(gh, N, B, out=Gct[:, slot * H:], out_stride=cs)
            gv = ops.add(gvc, ops.linear(gh, d["wxT"]))
        # dW1[l, b] = gt3^T t2, dW0[l, b] = gt2^T relu(h): two grouped launches over the tape tensors as they lie
        ops.conv_wgrad_batched(ft["t2"].view(L * NB, R, H), gt3_all.view(L * NB, R, H), raw(self.raw_w1, (H, H)), H * H, L * NB)
        ops.conv_wgrad_batched(ft["tb"].view(L * NB, R, H), gt2_all.view(L * NB, R, H), raw(self.raw_w0, (H, H)), H * H, L * NB)
        ops.colsum(bsum, raw(self.raw_bias, (bs_w,)))
        ops.linear_wgrad(tp["feat"], Gct, raw(self.raw_wctx, (cs, g.context_features))); ops.colsum(Gct, raw(self.raw_bctx, (cs,)))
        g_feat = ops.linear(Gct, self.wctxT)
        self._reparam_backward(g_logp)
        return g_feat

    def backward(self, g_x, g_logp, N, B):
        """g_x (R,45) = dL/d sample, g_logp (B,) = dL/d log_p per image (None: no entropy term).  Writes every Glow
        parameter's gradient into the trainer's raw arena and returns dL/d feat (B, F) through the context terms."""
        if self._tp.get("fused") is not None and self.mixed:
            if self._tp.get("chain"):
                return self._backward_chain(g_x, g_logp, N, B)
            return self._backward_fused(g_x, g_logp, N, B)
        ts, g = self.ts, self.g
        L_, D, H, R = _lib.lib(), g.features, g.hidden, g_x.shape[0]
        tp = self._tp
        s, dev = ops._stream, g_x.device
        raw = lambda o, shape: ts._raw(o, shape)
        ctab, cs = tp["ctab"], tp["ctab"].shape[1]
        gv = torch.empty(R, 64, device=dev)
        ops.check(L_.mhe_pad64_f32(ops._ptr(g_x), ops._ptr(gv), R, D, s()), "mhe_pad64_f32")
        Gct = torch.zeros(B, cs, device=dev)
        fused = tp.get("fused")
        for l in range(g.num_layers):
            d = rs = self.layers[l]
            if fused is not None:       # the one-launch kernel's tape: relu(h) of every block and the final layer's operand as bf16
                bits = tp["bits"]
                t = {"y": fused["y"][l], "v": fused["v"][l], "prm": fused["prm"][l], "hf": fused["hf"][l],
                     "tb": [fused["tb"][l, b_].view(R, 1, 1, H) for b_ in range(g.num_blocks)],
                     "t2": [fused["t2"][l, b_].view(R, 1, 1, H) for b_ in range(g.num_blocks)],
                     "t3": [fused["t3"][l, b_].view(R, 1, 1, H) for b_ in range(g.num_blocks)],
                     "drop": [None if bits is None else bits[l, b_] for b_ in range(g.num_blocks)]}
            else:
                t = tp["tape"][l]
            slot = l * self.per
            ops.linear_wgrad(t["y"], gv, raw(rs["r_ainv"], (64, 64))); ops.colsum(gv, raw(rs["r_cinv"], (64,)))
            gy = ops.linear(gv, self.aff["AinvT"][l])
            gvc, gprm = torch.empty(R, 64, device=dev), torch.empty(R, 64, device=dev)
            ops.check(L_.mhe_glow_coupling_inv_bwd_f32(ops._ptr(t["v"]), ops._ptr(t["prm"]), ops._ptr(gy), ops._ptr(g_logp), -1.0 / N,
                                                       ops._ptr(gvc), ops._ptr(gprm), R, B, D, d["first"], d["T"], s()), "mhe_glow_coupling_inv_bwd_f32")
            if fused is not None:       # the final layer's operand was kept as bf16: its weight gradient on bf16 operands, f32 accumulation
                ops.conv_wgrad(t["hf"].view(R, 1, 1, H), gprm.to(torch.bfloat16).view(R, 1, 1, 64), 1, 1, 1, 0, raw(rs["r_wf"], (64, H)))
            else:
                ops.linear_wgrad(t["hs"][-1], gprm, raw(rs["r_wf"], (64, H)))
            ops.colsum(gprm, raw(rs["r_bf"], (64,)))
            gh = ops.linear(gprm, d["wfT"])
            for b in range(g.num_blocks - 1, -1, -1):
                rb = rs["r_blocks"][b]
                w0T, w1T = d["blocksT"][b]
                t3b, t2b = t["t3"][b], t["t2"][b]
                gt3, ggate = torch.empty_like(t3b), torch.empty(R, H, device=dev)
                ops.check(L_.mhe_glow_glu_bwd_f32(ops._ptr(gh), ops._ptr(t3b), C.c_void_p(ctab[:, (slot + 1 + b) * H:].data_ptr()), cs,
                                                  ops._ptr(gt3), ops._ptr(ggate), R, H, 1, B, ops.dtype_code(t3b.dtype), s()), "mhe_glow_glu_bwd_f32")
                ops.sum_over_hypotheses(ggate, N, B, out=Gct[:, (slot + 1 + b) * H:], out_stride=cs)
                if self.mixed:
                    # the four h x h products of the block's reverse pass on bf16 MFMA; bias sums from an f32 view of the bf16 gradient
                    _, _, w0Tb, w1Tb = d["blocks_b"][b]
                    ops.conv_wgrad(t2b, gt3, 1, 1, 1, 0, raw(rb["w1"], (H, H))); ops.colsum(gt3, raw(rb["b1"], (H,)))
                    gt2 = ops.conv2d_nhwc(gt3, w1Tb, 1, 1, 1, 0)
                    if t["drop"][b] is not None:          # dropout's reverse: the same mask and scale on the gradient
                        ops.dropout_(gt2, g.p_drop, bits=t["drop"][b])
                    ops.flow_lrelu_bwd_mixed(gt2.view(R, H), t2b.view(R, H), out_bf16=gt2.view(R, H), slope=0.0)
                    if fused is not None:
                        tt = t["tb"][b]
                    else:
                        tt = torch.empty(R, 1, 1, H, device=dev, dtype=torch.bfloat16)
                        ops.check(L_.mhe_relu_copy_f32(ops._ptr(t["hs"][b]), ops._ptr(tt), tt.numel(), ops.BF16, s()), "mhe_relu_copy_f32")
                    ops.conv_wgrad(tt, gt2, 1, 1, 1, 0, raw(rb["w0"], (H, H))); ops.colsum(gt2, raw(rb["b0"], (H,)))
                    gt = ops.conv2d_nhwc(gt2, w0Tb, 1, 1, 1, 0)
                    if fused is not None:       # relu(h) > 0  <=>  h > 0: the gate from the kept bf16 activation
                        ops.check(L_.mhe_relu_bwd_add_mixed(ops._ptr(gh), ops._ptr(gt), ops._ptr(tt), gh.numel(), ops.BF16, ops.BF16, s()), "mhe_relu_bwd_add_mixed")
                    else:
                        ops.check(L_.mhe_relu_bwd_add_f32(ops._ptr(gh), ops._ptr(gt), ops._ptr(t["hs"][b]), gh.numel(), ops.BF16, s()), "mhe_relu_bwd_add_f32")
                    continue
                ops.linear_wgrad(t2b, gt3, raw(rb["w1"], (H, H))); ops.colsum(gt3, raw(rb["b1"], (H,)))
                gt2 = ops.linear(gt3, w1T)
                if t["drop"][b] is not None:
                    ops.dropout_(gt2, g.p_drop, bits=t["drop"][b])
                ops.flow_lrelu_bwd(gt2, t2b, slope=0.0)
                tt = torch.empty(R, H, device=dev)
                ops.check(L_.mhe_relu_copy_f32(ops._ptr(t["hs"][b]), ops._ptr(tt), tt.numel(), 0, s()), "mhe_relu_copy_f32")
                ops.linear_wgrad(tt, gt2, raw(rb["w0"], (H, H))); ops.colsum(gt2, raw(rb["b0"], (H,)))
                gt = ops.linear(gt2, w0T)
                ops.check(L_.mhe_relu_bwd_add_f32(ops._ptr(gh), ops._ptr(gt), ops._ptr(t["hs"][b]), gh.numel(), 0, s()), "mhe_relu_bwd_add_f32")
            ops.linear_wgrad(t["v"], gh, raw(rs["r_wx"], (H, 64)))
            ops.sum_over_hypotheses(gh, N, B, out=Gct[:, slot * H:], out_stride=cs)
            gv = ops.add(gvc, ops.linear(gh, d["wxT"]))
        ops.linear_wgrad(tp["feat"], Gct, raw(self.raw_wctx, (cs, g.context_features))); ops.colsum(Gct, raw(self.raw_bctx, (cs,)))
        g_feat = ops.linear(Gct, self.wctxT)
        self._reparam_backward(g_logp)
        return g_feat

    def _reparam_backward(self, g_logp):
        """ActNorm / LU parameter gradients from dAinv, dcinv (in the raw arena) and the log-det constant: mhe_glow_reparam_bwd_f64, one
        workgroup per layer, float64, written straight into the six raw-gradient slots of every layer"""
        ts = self.ts
        if self._gtabs is None or self._gtabs[0] != ts.raw.data_ptr():
            base = ts.raw.data_ptr()
            at = lambda o: base + 4 * o
            mk = lambda rows: torch.tensor(rows, dtype=torch.int64, device=ts.dev)
            self._gtabs = (base, mk([at(d["r_ainv"]) for d in self.layers]), mk([at(d["r_cinv"]) for d in self.layers]),
                           mk([[at(d["r_" + k]) for k in ("log_scale", "shift", "lower", "upper", "udiag", "bias")] for d in self.layers]))
        _, ga, gc, gp = self._gtabs
        # sum_r dL/dlog q[r] = -sum_b g_logp[b] (the entropy term: each image's K rows carry g_logp[b] * (-1 / K))
        ops.glow_reparam_bwd(ga, gc, g_logp, self.g.num_layers, self.g.features, self.aff["ws"], gp, q_sign=-1.0)
