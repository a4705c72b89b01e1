"""ConditionalGlow with the call surface the reference uses for `q_z_giv_i_model == 'glow'`
(reference hand/network.py:342-344 ctor `(45, 512, 4, 2, context_features=512, dropout_probability=0.2)`;
:693-694 `log_prob(z, context=feat) -> (log_prob, z)`; :736-742 `sample_and_log_prob(N, noise, context)
-> (samples (B,N,D), log_prob (B,N), z)`; `_distribution._shape`), on HIP kernels.

**PARITY UNPINNED.**  The reference imports the class from the ProHMR fork of nflows
(`git+https://github.com/nkolot/nflows.git`, unpinned, hand/environment.yml:284), which is neither vendored
nor installed; no reference test or fixture covers it.  This module follows the published nflows algorithm as
restated in oracle/glow_ref.py (per layer ActNorm -> LULinear -> AffineCouplingTransform whose scale/shift come
from a context-conditioned ResidualNet with GLU gating; alternating +-1 mask; StandardNormal base) and keeps
nflows' module tree so that its state_dict keys (`_transform._transforms.{i}...`) line up.  Dropout (p = 0.2 in the reference's
constructor call) is active in train mode as in the reference - masks drawn on the device (mhe_dropout; the reference's draws come from
torch's generator and cannot be reproduced, so parity tests record the masks and hand them to the oracle) - and the identity in eval mode;
batch norm inside the nets is off (nflows' default).

MI355X shape of the computation: ActNorm and the LU product collapse into one 45x45 affine map per layer (and its
inverse for sampling), the flow variable is carried zero-padded to 64 columns so every dense product is an
mhe_linear_f32 call, and everything that depends on the context only (initial-layer context columns, the GLU
gates of every block of every layer) is ONE GEMM per image, indexed per hypothesis row by the kernels.
"""
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops, _lib


class _ActNorm(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.register_buffer("initialized", torch.tensor(True))
        self.log_scale = nn.Parameter(torch.zeros(features))
        self.shift = nn.Parameter(torch.zeros(features))


class _LULinear(nn.Module):
    def __init__(self, features, eps=1e-3):
        super().__init__()
        self.features, self.eps = features, eps
        n = features * (features - 1) // 2
        self.lower_entries = nn.Parameter(torch.zeros(n))
        self.upper_entries = nn.Parameter(torch.zeros(n))
        self.unconstrained_upper_diag = nn.Parameter(torch.full((features,), math.log(math.exp(1 - eps) - 1)))   # identity init
        self.bias = nn.Parameter(torch.zeros(features))


    def weight_and_diag(self):
        """W = L U and U's diagonal in float64 on the host (the wide body flow, features > 64; the hand flow builds them on the device)"""
        D = self.features
        lower = torch.zeros(D, D, dtype=torch.float64)
        li = np.tril_indices(D, k=-1)
        lower[li[0], li[1]] = self.lower_entries.detach().double().cpu()
        lower[range(D), range(D)] = 1.0
        upper = torch.zeros(D, D, dtype=torch.float64)
        ui = np.triu_indices(D, k=1)
        upper[ui[0], ui[1]] = self.upper_entries.detach().double().cpu()
        diag = F.softplus(self.unconstrained_upper_diag.detach().double().cpu()) + self.eps
        upper[range(D), range(D)] = diag
        return lower @ upper, diag


class _ResidualBlock(nn.Module):
    def __init__(self, features, context_features, dropout_probability):
        super().__init__()
        self.context_layer = nn.Linear(context_features, features)
        self.linear_layers = nn.ModuleList([nn.Linear(features, features) for _ in range(2)])
        self.dropout = nn.Dropout(p=dropout_probability)
        nn.init.uniform_(self.linear_layers[-1].weight, -1e-3, 1e-3)     # nflows zero_initialization
        nn.init.uniform_(self.linear_layers[-1].bias, -1e-3, 1e-3)


class _ResidualNet(nn.Module):
    def __init__(self, in_features, out_features, hidden_features, context_features, num_blocks, dropout_probability):
        super().__init__()
        self.initial_layer = nn.Linear(in_features + context_features, hidden_features)
        self.blocks = nn.ModuleList([_ResidualBlock(hidden_features, context_features, dropout_probability) for _ in range(num_blocks)])
        self.final_layer = nn.Linear(hidden_features, out_features)


class _AffineCoupling(nn.Module):
    def __init__(self, mask, make_net):
        super().__init__()
        idx = torch.arange(len(mask))
        self.register_buffer("identity_features", idx[mask <= 0])
        self.register_buffer("transform_features", idx[mask > 0])
        self.transform_net = make_net(int((mask <= 0).sum()), 2 * int((mask > 0).sum()))


class _Composite(nn.Module):
    def __init__(self, transforms):
        super().__init__()
        self._transforms = nn.ModuleList(transforms)


class _StandardNormal(nn.Module):
    def __init__(self, shape):
        super().__init__()
        self._shape = torch.Size(shape)
        self.register_buffer("_log_z", torch.tensor(0.5 * np.prod(shape) * np.log(2 * np.pi), dtype=torch.float64), persistent=False)


class ConditionalGlow(nn.Module):
    def __init__(self, features, hidden_features, num_layers, num_blocks_per_layer, activation=F.relu, dropout_probability=0.5,
                 context_features=None, batch_norm_within_layers=False):
        super().__init__()
        if context_features is None or batch_norm_within_layers or activation is not F.relu:
            raise NotImplementedError("only the context-conditioned, batch-norm-free ReLU configuration the reference builds")
        if features > 256 or hidden_features % 64 or context_features % 64:
            raise NotImplementedError(f"unsupported geometry features={features} hidden={hidden_features} context={context_features}")
        # the flow variable / the nets' parameter rows are carried zero-padded to multiples of 64 columns
        # (45 -> 64 for the hand flow; 144 -> 192 for a ProHMR-style 24 x 6D body pose)
        self.Dp = (features + 63) // 64 * 64
        self.features, self.hidden, self.num_layers, self.num_blocks, self.context_features = \
            features, hidden_features, num_layers, num_blocks_per_layer, context_features
        mask = torch.ones(features)
        mask[::2] = -1
        make = lambda i, o: _ResidualNet(i, o, hidden_features, context_features, num_blocks_per_layer, dropout_probability)
        layers = []
        for _ in range(num_layers):
            layers += [_ActNorm(features), _LULinear(features), _AffineCoupling(mask.clone(), make)]
            mask = -mask
        self._transform = _Composite(layers)
        self._distribution = _StandardNormal([features])
        self._embedding_net = nn.Identity()
        self._pack = None
        # train-mode dropout of the residual blocks (nflows ResidualBlock: after the second activation; hand/network.py:343-344 builds the
        # flow with dropout_probability=0.2 and trains it that way, :781 "Since uses Dropout"): masks drawn on the device (ops.dropout_).
        # mask_feed: a list of mask-bit tensors consumed in call order instead of drawing (parity tests); record_masks: the bits of every
        # dropout of a pass are appended to last_masks (handed to the oracle, which cannot draw the same stream)
        self.p_drop = float(dropout_probability)
        self.mask_feed, self.record_masks, self.last_masks = None, False, []
        # operand dtype of the four hidden x hidden products per layer (95 % of the flow's FLOP): float32 (parity mode) or
        # bfloat16 with f32 accumulate (performance mode; forward / loss / sample only - the train step's pass stays f32)
        self.compute_dtype = torch.float32

    # ---- derived device operands, rebuilt when a parameter changes --------------------------------------
    def small_param_table(self):
        """int64 [layers, 6] device tensor: addresses of log_scale, shift, lower_entries, upper_entries, unconstrained_upper_diag, bias of
        every layer (the operand of ops.glow_affine); rebuilt when a parameter's storage moves"""
        T = self._transform._transforms
        rows = [[p.data_ptr() for p in (T[3 * l].log_scale, T[3 * l].shift, T[3 * l + 1].lower_entries, T[3 * l + 1].upper_entries,
                                        T[3 * l + 1].unconstrained_upper_diag, T[3 * l + 1].bias)] for l in range(self.num_layers)]
        key = tuple(map(tuple, rows))
        if getattr(self, "_ptab", None) is None or self._ptab[0] != key:
            self._ptab = (key, torch.tensor(rows, dtype=torch.int64, device=next(self.parameters()).device))
        return self._ptab[1]

    def _packed(self):
        ext = getattr(self, "_external_pack", None)
        if ext is not None:          # a train.TrainStep owns the parameters: its device-resident operand layouts, refreshed every step
            return ext()
        dev = next(self.parameters()).device
        ver = tuple(p._version for p in self.parameters()) + (str(dev),)
        if self._pack is not None and self._pack[0] == ver:
            return self._pack[1]
        D, H, Fc, T = self.features, self.hidden, self.context_features, self._transform._transforms
        if D <= 64:
            # ActNorm + LU of every layer as one 45x45 affine map and its inverse: one launch, float64 on the device (no host round trip)
            aff = ops.glow_affine(self.small_param_table(), self.num_layers, D, T[1].eps)
            pk = {"layers": [], "const_parts": aff["const_parts"], "aff": aff}
        else:
            # the wide body flow (144-D pose, padded to 192 columns): float64 on the host, once per parameter version
            pk = {"layers": [], "aff": None}
            parts = []
        wctx, bctx = [], []
        for l in range(self.num_layers):
            cp = T[3 * l + 2]
            if D <= 64:
                d = {"A": aff["A"][l], "c": aff["c"][l], "Ainv": aff["Ainv"][l], "cinv": aff["cinv"][l]}
            else:
                an, lu = T[3 * l], T[3 * l + 1]
                W, diag = lu.weight_and_diag()
                sc = torch.exp(an.log_scale.detach().double().cpu())
                A = W * sc[None, :]                                          # x -> W (s*x + shift) + b
                c = W @ an.shift.detach().double().cpu() + lu.bias.detach().double().cpu()
                Ainv = torch.linalg.inv(A)
                Dp = self.Dp
                pad = lambda M, v: (F.pad(M, (0, Dp - D, 0, Dp - D)).float().to(dev).contiguous(), F.pad(v, (0, Dp - D)).float().to(dev).contiguous())
                d = {}
                d["A"], d["c"] = pad(A, c)
                d["Ainv"], d["cinv"] = pad(Ainv, -(Ainv @ c))
                parts.append(float(an.log_scale.detach().double().sum().cpu() + torch.log(diag).sum()))
            net = cp.transform_net
            idf = cp.identity_features
            w0 = net.initial_layer.weight.detach()
            wx = torch.zeros(H, self.Dp, device=dev)
            wx[:, idf.to(dev)] = w0[:, :idf.numel()]
            d["wx"] = wx.contiguous()
            wctx.append(w0[:, idf.numel():]); bctx.append(net.initial_layer.bias.detach())
            d["blocks"] = []
            for blk in net.blocks:
                d["blocks"].append(tuple(t.detach().contiguous() for t in (blk.linear_layers[0].weight, blk.linear_layers[0].bias,
                                                                         blk.linear_layers[1].weight, blk.linear_layers[1].bias)))
                d.setdefault("blocks_bf16", []).append((blk.linear_layers[0].weight.detach().to(torch.bfloat16).contiguous(),
                                                        blk.linear_layers[1].weight.detach().to(torch.bfloat16).contiguous()))
                wctx.append(blk.context_layer.weight.detach()); bctx.append(blk.context_layer.bias.detach())
            nt = int(cp.transform_features.numel())
            Pp = (2 * nt + 63) // 64 * 64
            wf = torch.zeros(Pp, H, device=dev); wf[:2 * nt] = net.final_layer.weight.detach()
            bf = torch.zeros(Pp, device=dev); bf[:2 * nt] = net.final_layer.bias.detach()
            d["wf"], d["bf"], d["T"], d["first"] = wf.contiguous(), bf.contiguous(), nt, 1 - (l % 2)      # (the alternating mask: odd columns first)
            pk["layers"].append(d)
        if D > 64:
            pk["const_parts"] = torch.tensor(parts, dtype=torch.float32, device=dev)
        pk["wctx"], pk["bctx"] = torch.cat(wctx).contiguous(), torch.cat(bctx).contiguous()
        if H == 512 and self.num_blocks == 2 and D <= 48:
            # the one-launch kernel's operands (csrc/glow_fwd.hip): bf16 copies in MFMA fragment order, the final layer's rows at the flow
            # variable's own columns
            nets = [T[3 * l + 2].transform_net for l in range(self.num_layers)]
            st = lambda f: torch.stack([f(n) for n in nets])
            fp = ops.glow_fused_layout(torch.stack([d["wx"] for d in pk["layers"]]),
                                       st(lambda n: torch.stack([b.linear_layers[0].weight.detach() for b in n.blocks])),
                                       st(lambda n: torch.stack([b.linear_layers[1].weight.detach() for b in n.blocks])),
                                       torch.stack([d["wf"] for d in pk["layers"]]), torch.stack([d["bf"] for d in pk["layers"]]),
                                       st(lambda n: torch.stack([b.linear_layers[0].bias.detach() for b in n.blocks])),
                                       st(lambda n: torch.stack([b.linear_layers[1].bias.detach() for b in n.blocks])), D)
            pk["fused"] = {k: (v.to(torch.bfloat16) if k.endswith("F") else v.float()).contiguous() for k, v in fp.items() if not k.endswith("T")}
        self._pack = (ver, pk)
        return pk

    def _drop_bits(self, R):
        """the dropout masks of one sampling pass of the one-launch kernel: uint8 [L, 2, R * 64] (layer, block; ops.dropout_'s bit format over
        [R, hidden]) or None (eval mode / p = 0).  mask_feed / record_masks speak the layer-by-layer path's CALL ORDER - layers L-1 .. 0, blocks
        0 .. 1 - so that tests written against it feed and read the same lists"""
        if not (self.training and self.p_drop > 0.0):
            return None
        L, NB = self.num_layers, self.num_blocks
        if self.mask_feed:
            fed = [self.mask_feed.pop(0) for _ in range(L * NB)]
            bits = torch.stack([torch.stack([fed[(L - 1 - l) * NB + b].reshape(-1) for b in range(NB)]) for l in range(L)]).contiguous()
        else:
            bits = ops.dropout_bits(L * NB * R * self.hidden, self.p_drop, next(self.parameters()).device).view(L, NB, R * self.hidden // 8)
        if self.record_masks:
            self.last_masks += [bits[L - 1 - k // NB, k % NB] for k in range(L * NB)]
        return bits

    def dropout_(self, t):
        """the residual block's dropout on its second activation `t`, in place (train mode only); returns the mask bits or None"""
        if not (self.training and self.p_drop > 0.0):
            return None
        given = self.mask_feed.pop(0) if self.mask_feed else None
        bits = ops.dropout_(t, self.p_drop, bits=given)
        if self.record_masks:
            self.last_masks.append(bits)
        return bits

    # ---- the two directions ------------------------------------------------------------------------------
    def _net(self, d, v, ctab, slot, R, row_div, n_img, bufs):
        """coupling parameters [R,64] of layer `d` from the (padded) variable v whose identity columns are current"""
        L, H = _lib.lib(), self.hidden
        h, t, t2 = bufs
        s = ops._stream
        ops.linear(v, d["wx"], out=h)
        cs = ctab.shape[1]
        ops.check(L.mhe_glow_add_image_rows_f32(ops._ptr(h), C.c_void_p(ctab[:, slot * H:].data_ptr()), cs, R, H, row_div, n_img, s()), "mhe_glow_add_image_rows_f32")
        bf16 = self.compute_dtype == torch.bfloat16 and H % 64 == 0
        if bf16:
            tb, t2b = (torch.empty(R, 1, 1, H, device=v.device, dtype=torch.bfloat16) for _ in range(2))
        for b, (w0, b0, w1, b1) in enumerate(d["blocks"]):
            if bf16:        # relu(h) -> bf16, two h x h products on bf16 MFMA (bias + relu in the kernel's epilogue), gate in f32
                w0b, w1b = d["blocks_bf16"][b]
                ops.check(L.mhe_relu_copy_f32(ops._ptr(h), ops._ptr(tb), h.numel(), ops.BF16, s()), "mhe_relu_copy_f32")
                ops.conv2d_nhwc(tb, w0b, 1, 1, 1, 0, out_shift=b0, relu_out=True, out=t2b)
                self.dropout_(t2b)
                t = ops.conv2d_nhwc(t2b, w1b, 1, 1, 1, 0, out_shift=b1, out=tb).view(R, H)
            else:
                ops.check(L.mhe_relu_copy_f32(ops._ptr(h), ops._ptr(t), h.numel(), ops.dtype_code(t.dtype), s()), "mhe_relu_copy_f32")
                ops.linear(t, w0, b0, relu=True, out=t2)
                self.dropout_(t2)
                ops.linear(t2, w1, b1, out=t)
            ops.check(L.mhe_glow_glu_residual_f32(ops._ptr(h), ops._ptr(t), ops.dtype_code(t.dtype), C.c_void_p(ctab[:, (slot + 1 + b) * H:].data_ptr()), cs, R, H, row_div,
                                                  n_img, s()), "mhe_glow_glu_residual_f32")
        return ops.linear(h, d["wf"], d["bf"])

    def _run(self, v_in, context, inverse, row_div, n_img):
        """v_in (R,D) data (forward) or noise (inverse); returns (out (R,D), log_prob (R,))"""
        ops._chk(v_in, torch.float32, "glow.in"); ops._chk(context, torch.float32, "glow.context", (context.shape[0], self.context_features))
        pk, L, D, H = self._packed(), _lib.lib(), self.features, self.hidden
        R = v_in.shape[0]
        dev = v_in.device
        s = ops._stream
        ctab = ops.linear(context, pk["wctx"], pk["bctx"])                       # every context-only term, once per image
        N = R // n_img
        if (inverse and self.compute_dtype == torch.bfloat16 and pk.get("fused") is not None and row_div in (1, N)
                and os.environ.get("MHE_GLOW_FUSED", "1") == "1" and ops.glow_layers_supported(N, n_img, D, H, self.num_layers, self.num_blocks)):
            # the sampling direction of all layers in ONE launch (csrc/glow_fwd.hip; bf16 operands on the products, f32 residual stream,
            # flow variable, coupling and affine map)
            rn, rb = (n_img, 1) if row_div == 1 else (1, N)
            return ops.glow_layers(v_in, ctab, pk["fused"], pk["aff"], self._drop_bits(R), self.p_drop, N, n_img, D, rn, rb)
        v = torch.empty(R, self.Dp, device=dev)
        ops.check(L.mhe_pad64_f32(ops._ptr(v_in), ops._ptr(v), R, D, s()), "mhe_pad64_f32")
        z_in = v
        logdet = torch.zeros(R, device=dev)
        bufs = tuple(torch.empty(R, H, device=dev) for _ in range(3))
        per = 1 + self.num_blocks
        order = range(self.num_layers - 1, -1, -1) if inverse else range(self.num_layers)
        for l in order:
            d = pk["layers"][l]
            if not inverse:
                v = ops.linear(v, d["A"], d["c"])
            prm = self._net(d, v, ctab, l * per, R, row_div, n_img, bufs)
            y = torch.empty(R, self.Dp, device=dev)
            ops.check(L.mhe_glow_coupling_f32(ops._ptr(v), ops._ptr(prm), ops._ptr(y), ops._ptr(logdet), R, D, d["first"], d["T"], int(inverse), s()),
                      "mhe_glow_coupling_f32")
            v = ops.linear(y, d["Ainv"], d["cinv"]) if inverse else y
        z = z_in if inverse else v                                               # the base-density argument
        return ops.glow_finish(z, v, logdet, R, D, inverse, pk["const_parts"])

    # ---- reference call surface ----------------------------------------------------------------------------
    def log_prob(self, inputs, context=None, rows_per_context=None):
        """(log_prob (R,), noise (R,D)).  `context` has R rows (the reference passes `feat.repeat(N,1)`), or B rows with
        sample-major inputs (row r uses context[r % B]; extension, hoists the context terms)."""
        R, Bc = inputs.shape[0], context.shape[0]
        if R % Bc:
            raise ValueError(f"glow rows ({R}) must be a multiple of context rows ({Bc})")
        z, lp = self._run(inputs.contiguous(), context.contiguous(), False, 1, Bc)
        return lp, z

    def sample_and_log_prob(self, num_samples, noise=None, context=None):
        """samples (B,N,D), log_prob (B,N), noise - rows batch-major as nflows lays them out."""
        B = context.shape[0]
        if noise is None:
            noise = ops.randn(B * num_samples, self.features, context.device).view(B, num_samples, self.features)      # drawn on the device (mhe_randn_f32)
        x, lp = self._run(noise.reshape(B * num_samples, self.features).contiguous(), context.contiguous(), True, num_samples, B)
        return x.view(B, num_samples, -1), lp.view(B, num_samples), noise

    def forward(self, context, num_samples=1):
        """ProHMR call form `flow(conditioning_feats, num_samples)` (reference README.md:34,39)"""
        s, lp, _ = self.sample_and_log_prob(num_samples, context=context)
        return s, lp
