"""Conditional RealNVP with the reference's module API and state_dict layout
(reference hand/flows.py:75-122 `_nets`, :125-362 `RealNVP`), evaluated by the
fused HIP coupling kernel (csrc/flow.hip).

Supported configuration = the one MHEnt builds (reference
hand/CrossModalHand.py:67-69): dim > 3, integer `tsfm_on` (feature-conditioned),
kemb=False, two hidden layers of equal width.  The per-joint / RLE experiment
branches of the reference (dim in {2,3}, kemb, tsfm_on in {'x','z'}) are dead with
the shipped config and raise NotImplementedError.

Difference by design: `cond` may have B rows while the flow variable has R = N*B
sample-major rows (row r uses cond[r % B]); with R rows of cond it is exactly
the reference's call (`feat.repeat(N,1)`, hand/network.py:734).  The
conditioning projections c0/c1 are then evaluated once per image.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops


class _nets(nn.Module):
    """parameter holder with the reference's names: l.{0,1,2}, c.{0,1}"""
    def __init__(self, dim, cond_dim=0, h_dims=(64, 64), s=True):
        super().__init__()
        self.cond_dim, self.s = cond_dim, s
        self.l = nn.ModuleList([nn.Linear(dim, h_dims[0]), nn.Linear(h_dims[0], h_dims[1]), nn.Linear(h_dims[1], dim)])
        if cond_dim:
            self.c = nn.ModuleList([nn.Linear(cond_dim, h) for h in h_dims])


class RealNVP(nn.Module):
    def __init__(self, nets=_nets, nett=_nets, mask=None, prior=None, dim=63, tsfm_on=None, kemb=False, jointN=21,
                 h_dims=(64, 64), num_steps=3, cond_mapping_dims=None):
        super().__init__()
        if dim <= 3 or kemb or not isinstance(tsfm_on, int) or cond_mapping_dims:
            raise NotImplementedError("only the feature-conditioned joint flow MHEnt uses is built "
                                      "(dim > 3, integer tsfm_on, kemb=False)")
        if len(h_dims) != 2 or h_dims[0] != h_dims[1] or h_dims[0] % 64 or dim > 48:
            raise NotImplementedError(f"unsupported geometry dim={dim}, h_dims={list(h_dims)}")
        self.dim, self.jointN, self.tsfm_on, self.hidden = dim, jointN, tsfm_on, h_dims[0]
        if mask is None:                       # reference hand/flows.py:153-155
            A = [0] * (dim // 2) + [1] * (dim - dim // 2)
            Bm = [1 - a for a in A]
            mask = torch.from_numpy(np.array([A, Bm] * num_steps).astype(np.float32))
        self.register_buffer("mask", mask)
        self.t = nn.ModuleList([nett(dim, cond_dim=tsfm_on, h_dims=h_dims, s=False) for _ in range(len(mask))])
        self.s = nn.ModuleList([nets(dim, cond_dim=tsfm_on, h_dims=h_dims) for _ in range(len(mask))])
        self.scale = 1.0
        self._pack = None
        # operand dtype of the coupling MLPs: float32 (parity mode) or bfloat16 (performance mode, f32 accumulate)
        self.compute_dtype = torch.float32

    # ---- device-side packed parameters, rebuilt when any weight changes ------
    def _packed(self):
        ext = getattr(self, "_external_pack", None)
        if ext is not None:         # kept current on the device by train.TrainStep (gathers from its flat parameter buffer)
            self._external_sync()   # ... re-gathered here if the parameters were written since (optimizer.step, load_state_dict)
            return ext
        bf16 = self.compute_dtype == torch.bfloat16 and self.hidden % 128 == 0
        ver = tuple(p._version for p in self.parameters()) + (str(self.mask.device), bf16)
        if self._pack is None or self._pack[0] != ver:
            dev = self.mask.device
            packs, b2, wc, bc = [], [], [], []
            for i in range(len(self.mask)):
                for net in (self.s[i], self.t[i]):
                    w = [l.weight.detach().cpu().numpy() for l in net.l]
                    packs.append((ops.flow_pack_net_bf16 if bf16 else ops.flow_pack_net)(w[0], w[1], w[2]))
                    b2.append(net.l[2].bias.detach())
                    for j in range(2):
                        wc.append(net.c[j].weight.detach())
                        bc.append(net.c[j].bias.detach() + net.l[j].bias.detach())
            stream = np.concatenate(packs)
            if bf16:
                stream = stream.view(np.int16)
            b2 = torch.stack(b2)
            if bf16:                      # the bf16 kernel wants l2.bias zero-padded to 64 per net
                b2 = torch.nn.functional.pad(b2, (0, 64 - b2.shape[1]))
            wc = torch.cat(wc).contiguous()
            fragp = None
            if bf16 and self.hidden == 512:       # the same weights in MFMA fragment order: operands of the fragment-streaming kernel
                per = []
                for i in range(len(self.mask)):
                    for net in (self.s[i], self.t[i]):
                        per.append(torch.cat([f.reshape(-1) for f in ops.flow_frag_pack(*(l.weight.detach() for l in net.l))]))
                fp = torch.stack(per).to(torch.bfloat16).to(dev).contiguous()              # [nets][W1 | W0 | W2 fragments]
                hh = self.hidden
                fragp = (fp[0, hh * hh:], fp[0], fp[0, hh * hh + 64 * hh:], fp.shape[1], fp)   # net 0's w0F, w1F, w2F, the net pitch (+ owner)
            self._pack = (ver, torch.from_numpy(stream).to(dev), b2.contiguous(), wc, torch.cat(bc).contiguous(),
                          wc.to(torch.bfloat16) if bf16 and wc.shape[1] % 64 == 0 else None, fragp)
        return self._pack[1:]

    def _cond_table(self, cond):
        """(B, F) features -> (B, 2*ncoup, 2, hidden): c_j(feat) + c_j.bias + l_j.bias per net.  In the bf16 mode the product takes
        bf16 operands like the nets' own layers do (f32 accumulation and bias; 23 us against 117 us for the f32 form at C2)."""
        _, _, wc, bc, wcb = self._packed()[:5]
        if wcb is not None:
            cb = getattr(cond, "_mhe_bf16", None)            # left by BasicEnc's l1 launch (same values, no cast launch)
            if cb is None or cb.shape != cond.shape:
                cb = cond.to(torch.bfloat16).contiguous()
            t = ops.linear_bf16_f32out(cb, wcb, bc)
        else:
            t = ops.linear(cond.contiguous(), wc, bc)
        return t.view(cond.shape[0], 2 * len(self.mask), 2, self.hidden)

    def _run(self, v, cond, direction):
        if cond is None:
            raise NotImplementedError("unconditional flow is not part of the hot path")
        R, B = v.shape[0], cond.shape[0]
        if R % B:
            raise ValueError(f"flow rows ({R}) must be a multiple of conditioning rows ({B})")
        pk = self._packed()
        wstream, b2 = pk[:2]
        fragp = pk[5] if len(pk) > 5 else None
        if (fragp is not None and os.environ.get("MHE_FLOW_FRAG", "1") == "1"
                and ops.flow_couplings_frag_supported(R, B, v.shape[1], self.hidden, len(self.mask))):
            return ops.flow_couplings_frag(v.contiguous(), self._cond_table(cond), fragp[0], fragp[1], fragp[2], fragp[3], b2, self.mask, B,
                                           self.hidden, direction)
        return ops.flow_couplings(v.contiguous(), self._cond_table(cond), wstream, b2, self.mask, B, self.hidden, direction)

    # ---- reference call surface ------------------------------------------------
    def forward_p(self, z, cond=None):
        """z -> x.  reference hand/flows.py:210-217"""
        return self._run(z, cond, ops.FLOW_FORWARD)[0]

    def backward_p(self, x, cond=None):
        """x -> (z, log_det_J).  reference hand/flows.py:219-227"""
        z, sum_s, _ = self._run(x, cond, ops.FLOW_INVERSE)
        return z, -sum_s

    def make_cond(self, feat):
        """reference hand/flows.py:229-269: identity for dim > 3 without a partitioner."""
        return feat

    def log_prob(self, x, mu=None, logvar=None, return_dict=False, weights=None, return_z=False):
        """reference hand/flows.py:271-331; `logvar` carries the image feature.
        Visibility weights other than all-ones are rejected like the reference does for dim > 3."""
        if weights is not None and bool((weights != 1).any()):
            raise NotImplementedError
        bs = x.shape[0]
        z, _, lp = self._run(x.reshape(-1, self.dim) / self.scale, self.make_cond(logvar), ops.FLOW_INVERSE)
        loss = lp.view(bs, -1).sum(1)
        if return_z:
            return z, loss
        return {"loss": loss} if return_dict else loss

    def sample(self, batchSize, temp=0.7, mu=None, logvar=None, return_z=False, noise=None):
        """reference hand/flows.py:333-359.  `noise` (batchSize, dim) ~ N(0,I) may be
        supplied for reproducible parity runs; otherwise it
        is drawn on the device."""
        bs = batchSize          # == logvar.shape[0] in the reference's own call (network.py:733-735)
        if noise is None:
            z0 = ops.randn(batchSize, self.dim, logvar.device, scale=temp)       # drawn on the device (mhe_randn_f32)
        else:
            z0 = (noise * temp).contiguous()
        x = self.forward_p(z0, cond=self.make_cond(logvar)) * self.scale
        if return_z:
            return x.view(bs, -1), z0.view(bs, -1)
        return x.view(bs, -1)

    def sample_with_log_prob(self, z0, cond):
        """One pass: x = forward_p(z0) and log q(x) = logN(z0) - sum s (SURVEY.md A2 ii)."""
        x, _, lq = self._run(z0, cond, ops.FLOW_FORWARD)
        return x, lq

    def forward(self, x):
        return self.log_prob(x)
