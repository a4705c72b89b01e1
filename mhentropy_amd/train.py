"""The reference's train step on HIP kernels, with an explicit reverse pass.

Reference (hand/CrossModalHand.py:455-470, optimizer :191-203):
    output = model(image, target); total_loss, losses, metrics = criterion(output, target)
    optimizer.zero_grad(); total_loss.backward()
    clip_grad_norm_(encoderRGB.parameters(), 1.);  optimizer.step()        # torch.optim.Adam(lr)
with total_loss = mean_b(-log_p[b]) (hand/criteria.py:55,173).

There is no autograd graph here: `TrainStep` runs the forward of MHEnt.get_loss stage by stage keeping
what the reverse pass needs (raw convolution outputs, post-activation tensors, BatchNorm batch
statistics, the flow sample), then walks the stages backwards through hand-written reverse kernels:
    loss -> MANO likelihood (mhe_mano_joints_bwd_f32) -> RealNVP couplings (re-evaluated layer by layer,
    csrc/flow_bwd.hip + mhe_conv_wgrad_nhwc) -> conditioning / det-head / l1 dense layers -> ResNet trunk
    (data gradients through the forward implicit-GEMM kernel on transposed, tap-flipped weights; weight
    gradients through mhe_conv_wgrad_nhwc; train-mode BatchNorm reverse) -> clip + Adam (one fused pass).

Memory plan (sized for 288 GB HBM): all parameters live in ONE flat f32 buffer (the nn.Parameters of the
model are views into it, so state_dict / eval paths are unchanged), with flat gradient and Adam moment
buffers of the same length - one RCCL all-reduce, one optimizer launch.  Every derived operand
layout (packed forward weights, dgrad operands, padded/transposed dense weights) is refreshed once per step
by a gather over an index table built once on the host; weight gradients land in a raw arena in whatever
layout their kernel produces and one final gather maps them onto the flat gradient buffer.
"""
import math

import os

import numpy as np
import torch

from . import ops, resnet
from .resnet import BN_EPS, BN_MOMENTUM


def _ceil(a, b):
    return (a + b - 1) // b * b


def dgrad_operand_index(idx):
    """index table of the data-gradient convolution's weight operand from the index tensor of a torch conv
    weight [Cout,Cin,KH,KW]:  W'[ci][kh'][kw'][co] = W[co][ci][KH-1-kh'][KW-1-kw'], rows = Cin, k = (kh',kw',co)"""
    Cout, Cin, KH, KW = idx.shape
    return idx.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, KH * KW * Cout)


def dgrad_s2_operand_indices(idx):
    """the four parity-class operands of a 3x3 / stride-2 / pad-1 convolution's data gradient (mhe_conv3x3s2_dgrad_nhwc) as index
    tables into the torch weight [Cout,Cin,3,3]: class (py, px) -> [Cin][(th, tw, co)] with forward taps kh = (1,) for py = 0 and
    (2, 0) for py = 1 (output row 2i+py reads gy rows i + th), likewise kw"""
    Cout, Cin, KH, KW = idx.shape
    assert KH == 3 and KW == 3
    taps = ((1,), (2, 0))
    out = []
    for py in range(2):
        for px in range(2):
            sub = idx[:, :, list(taps[py])][:, :, :, list(taps[px])]            # [Cout, Cin, th, tw]
            out.append(sub.permute(1, 2, 3, 0).reshape(Cin, len(taps[py]) * len(taps[px]) * Cout))
    return out


def conv_dgrad(gy, w_dg, k, stride, pad, H, W, residual=None, mask=None, bn=None, w_s2=None, res_half=False, coarse=False, mask_bits=None):
    """gradient of a k x k / stride / pad convolution w.r.t. its [B,H,W,Cin] input (+ residual), through the
    FORWARD implicit-GEMM kernel: stride 1 is a convolution of gy with the transposed, tap-flipped weights at
    padding k-1-pad; a stride-2 3x3 runs the same on the zero-dilated gy; a stride-2 1x1 is computed on the
    coarse grid and scattered to the even positions.  mask (the convolution's own post-ReLU input): the result is
    gated by [mask > 0] in the kernel's epilogue, i.e. it leaves as the gradient w.r.t. the PRE-activation."""
    if stride == 1:
        return ops.conv2d_nhwc(gy, w_dg, k, k, 1, k - 1 - pad, residual=residual, mask=mask, bn=bn, res_half=res_half,
                               mask_bits=mask_bits if mask is not None else None)
    if stride != 2 or k not in (1, 3):
        raise NotImplementedError(f"conv_dgrad: k={k} stride={stride}")
    if k == 3 and w_s2 is not None and pad == 1 and H == 2 * gy.shape[1] and W == 2 * gy.shape[2]:
        return ops.conv3x3s2_dgrad(gy, w_s2, residual=residual, mask=mask, bn=bn)       # four parity classes, no zero-dilated copy
    if k == 3:
        return ops.conv2d_nhwc(ops.upsample2(gy, H, W), w_dg, 3, 3, 1, 1, residual=residual, mask=mask, bn=bn)
    if mask is not None:
        raise NotImplementedError("conv_dgrad: mask with a stride-2 1x1 (only the un-gated downsample branch uses it)")
    half = ops.conv2d_nhwc(gy, w_dg, 1, 1, 1, 0)
    if coarse and residual is None:     # the consumer adds it at the even positions itself (mhe_conv_desc.res_half): no scattered copy
        return half
    return ops.upsample2(half, H, W, base=residual)


def halo_operand_index(f):
    """standard 3x3 pack [Cout][9 Cin] (index table) -> the fragment-major stages of csrc/conv_halo.hip (ops.conv3x3_halo_pack on indices):
    [Cout / 128][Cin / 64][9][4 channel tiles x 4 k-steps][2 k halves x 32 rows][8]"""
    Cout, K = f.shape
    Cin = K // 9
    v = f.view(Cout // 128, 4, 32, 9, Cin // 64, 4, 2, 8)          # ntile, nt32, row, tap, chunk, ks, kg, e
    return v.permute(0, 4, 3, 1, 5, 6, 2, 7).reshape(Cout, K).contiguous()


def flow_stream_table(dim, h, bf16):
    """gather table of ONE net's fragment-ordered weight stream, as local indices into [W0 | W1 | W2]
    (obtained by running the host packer on index-valued weights)"""
    n0, n1, n2 = h * dim, h * h, dim * h
    loc = np.arange(n0 + n1 + n2, dtype=np.int64)
    parts = lambda a: (a[:n0].reshape(h, dim), a[n0:n0 + n1].reshape(h, h), a[n0 + n1:].reshape(dim, h))
    if not bf16:
        w = parts((loc + 1).astype(np.float32))               # < 2^24: exact in f32
        return ops.flow_pack_net(*w).astype(np.int64) - 1
    out, pad = None, None
    for dig in range(3):                                        # base-128 digits (+1) are exact in bf16
        w = parts((((loc >> (7 * dig)) & 127) + 1).astype(np.float32))
        s = ops.flow_pack_net_bf16(*w)
        v = (s.astype(np.uint32) << 16).view(np.float32).astype(np.int64)
        if dig == 0:
            pad, out = v == 0, np.zeros_like(v)
        out += (np.maximum(v, 1) - 1) << (7 * dig)
    out[pad] = -1
    return out



class _Unit:
    """one convolution + BatchNorm of the trunk"""
    def __init__(self, conv, bn, k, stride, pad):
        self.conv, self.bn, self.k, self.stride, self.pad = conv, bn, k, stride, pad
        self.cin, self.cout = conv.in_channels, conv.out_channels


class TrainStep:
    def __init__(self, model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, dist=None, shard_hypotheses=False):
        self.model, self.lr, self.betas, self.eps, self.max_norm, self.dist = model, lr, betas, eps, max_norm, dist
        self.world = dist.get_world_size() if dist is not None else 1
        from .dist import forced
        # collectives are issued: more than one rank, or a forced group of ONE (the one-GPU rehearsal of the RCCL path, MHE_DIST_FORCE=1)
        self.comm = dist is not None and (self.world > 1 or forced())
        # hypothesis-sharded exchange (dist.HypothesisShards): images stay sharded for the encoder, every rank evaluates its
        # slice of the K hypotheses for ALL images from the gathered conditioning features
        self.shard_hypotheses = bool(shard_hypotheses) and self.comm
        trunk = model.feat_extractor.res
        self.trunk = trunk
        self.T = trunk.compute_dtype
        self.dev = next(model.parameters()).device
        if self.dev.type != "cuda":
            raise RuntimeError("TrainStep needs the model on a HIP device (there is no CPU path)")
        self._flatten_params()
        import os
        # derived operand layouts: two arenas (f32 / bf16), each refreshed by ONE gather per step over one concatenated index
        # table (a gather per tensor was ~340 launches of ~5 us)
        # ... and a second pair for the operand layouts only the FALLBACK paths of the flow read (coupling-by-coupling reverse pass, the
        # second-generation coupling kernel for hypothesis counts that are not a multiple of 64 per image): refreshed lazily, when such a path
        # runs (`_need_fallback`) - with the one-launch forward and reverse kernels they were gathered every step to be read by nobody
        # (33 M of the 118 M+ gathered elements at the shipped flow's size)
        self._arena = {dt: {"buf": torch.zeros(4 * self.n_params, device=self.dev, dtype=dt), "used": 0, "idx": [], "idx2": []}
                       for dt in (torch.float32, torch.bfloat16)}
        self._arena_fb = {dt: {"buf": None, "used": 0, "idx": [], "idx2": []} for dt in (torch.float32, torch.bfloat16)}
        self._fb_lazy = os.environ.get("MHE_LAZY_FALLBACK_TABLES", "1") == "1"
        self._fb_keep = not self._fb_lazy     # True: every repack refreshes them (as soon as a step has needed them once)
        self._fb_stale = False
        self._poison_stale = os.environ.get("MHE_POISON_STALE_TABLES", "0") == "1"
        self._raw_n = 0
        self._unpack = torch.full((self.n_params,), -1, dtype=torch.int64)
        self.cond_bwd_bf16 = os.environ.get("MHE_COND_BWD_BF16", "1") == "1"      # (read by _build_flow)
        self._build_trunk()
        self._build_heads()
        self._build_flow()
        # BatchNorm-reverse sums accumulated by the data-gradient epilogues (no separate reduce pass); MHE_BN_REDUCE_FUSED=0: separate pass
        self.fuse_bn_reduce = os.environ.get("MHE_BN_REDUCE_FUSED", "1") == "1"
        self.train_recompute = os.environ.get("MHE_TRAIN_RECOMPUTE", "1") == "1"
        self.conv3_fold = os.environ.get("MHE_CONV3_FOLD", "1") == "1"
        self.conv3_fold_cat = os.environ.get("MHE_CONV3_FOLD_CAT", "1") == "1"
        self.stem_bwd_two_pass = os.environ.get("MHE_STEM_BWD_TWO_PASS", "1") == "1"
        self.gate_bits = os.environ.get("MHE_GATE_BITS", "1") == "1"
        # 3x3 / stride-1 units of layer2 / layer3: the resident-tile kernel (csrc/conv_halo.hip) - forward with conv1's BatchNorm + ReLU on its
        # load (the normalised tensor written once on the way, no bn_act pass); their data gradients too (MHE_CONV_HALO_DG=0: the im2col kernels)
        self.shortcut_fold = os.environ.get("MHE_SHORTCUT_FOLD", "1") == "1"
        self.conv_halo = os.environ.get("MHE_CONV_HALO", "1") == "1"
        self.conv_halo_dg = os.environ.get("MHE_CONV_HALO_DG", "1") == "1"
        # (... with their own BatchNorm reverse on that launch's operand load, MHE_HALO_BN_ON_LOAD=1: built, equal - 26.35 / 26.38 ms: the transfer
        # waves' second operand stream and arithmetic cost the launch what the apply pass cost)
        self.halo_bn_on_load = os.environ.get("MHE_HALO_BN_ON_LOAD", "0") == "1"
        # the stem's BatchNorm + ReLU folded into its max pool, forward and reverse (ops.maxpool3x3s2_idx / maxpool3x3s2_bwd_bn)
        self.stem_pool_fused = os.environ.get("MHE_STEM_POOL_FUSED", "1") == "1"
        # the stem's BatchNorm-reverse sums from the pooled tensors (+ the raw winners kept by the forward's pool) instead of a first walk over
        # its full-resolution output: MHE_STEM_POOLED_SUMS=0 restores the walk
        self.stem_pooled_sums = os.environ.get("MHE_STEM_POOLED_SUMS", "1") == "1"
        self.bn_on_load_wide = os.environ.get("MHE_BN_BWD_ON_LOAD_WIDE", "1") == "1"
        # the trunk's weight gradients queued per gradient bucket and launched together (MHE_WGRAD_MULTI=0: one launch per layer)
        self.wgrad_multi = os.environ.get("MHE_WGRAD_MULTI", "1") == "1"
        self._wq = []
        self._bucket_bounds = self._gradient_buckets()
        self._xchg = None
        self.raw = torch.zeros(self._raw_n, device=self.dev, dtype=torch.float32)
        for u in self._raw_views:
            u()
        self._unpack_idx = self._unpack.to(torch.int32).to(self.dev)
        for a in list(self._arena.values()) + list(self._arena_fb.values()):
            n = a["used"]
            if a["buf"] is None:
                a["buf"] = torch.zeros(0, device=self.dev, dtype=torch.float32)
            a["view"] = a["buf"][:n]
            i1 = torch.cat(a["idx"]) if a["idx"] else torch.zeros(0, dtype=torch.int64)
            a["idx"] = i1.to(torch.int32).to(self.dev).contiguous()
            i2 = torch.cat(a["idx2"]) if a["idx2"] else torch.zeros(0, dtype=torch.int64)
            a["idx2"] = i2.to(torch.int32).to(self.dev).contiguous() if bool((i2 >= 0).any()) else None
            # the bf16 operand layouts as (base, stride, validity) per eight elements instead of eight indices (472 -> 133 MB of index reads per
            # step at C2) - when every group of the arena is affine in the parameter vector (permuted / padded weight tensors are)
            a["aff"] = None
            if (a["buf"].dtype == torch.bfloat16 and a["idx2"] is None and i1.numel() and i1.numel() % 8 == 0
                    and os.environ.get("MHE_GATHER_AFFINE", "1") == "1"):
                bs, msk, bad = ops.affine8(i1, with_bad=True)
                # groups that are not affine (the stem's 7 x 7 x 3 taps: 1.3 k of 10.6 M groups at ResNet-50) stay on the indexed form: a few
                # ranges of groups, merged when less than 4,096 groups apart
                ranges = []
                for g in bad.nonzero().flatten().tolist():
                    if ranges and g - ranges[-1][1] < 4096:
                        ranges[-1][1] = g + 1
                    else:
                        ranges.append([g, g + 1])
                if len(ranges) <= 4 and sum(r[1] - r[0] for r in ranges) * 50 < bad.numel():
                    segs, at = [], 0
                    for lo, hi in ranges + [[bad.numel(), bad.numel()]]:
                        if lo > at:
                            segs.append(("aff", at, lo))
                        if hi > lo:
                            segs.append(("idx", lo, hi))
                        at = hi
                    a["aff"] = (bs.to(self.dev), msk.to(self.dev), segs)
        self.step_t = torch.zeros(1, device=self.dev, dtype=torch.int32)
        self._zeros_c = torch.zeros(4096, device=self.dev, dtype=torch.float32)
        self.sq = torch.zeros(1, device=self.dev, dtype=torch.float32)
        self._ws = {}
        # the modules' forward paths (eval, sample) read the same device-resident operand packs, refreshed after every
        # optimizer step - no host re-pack, never stale
        self.repack()
        if self.glow is None:
            f0 = self.fnets[0]
            fragp = ((f0["f0F"], f0["f1F"], f0["f2F"], (self.fnets[1]["f1F"].data_ptr() - f0["f1F"].data_ptr()) // 2)
                     if "f1F" in f0 and len(self.fnets) > 1 else None)
            self.flow._external_pack = (self.f_stream, self.f_b2, self.f_wc, self.f_bc, self.f_wcb, fragp)
            self.flow._external_sync = self.sync_all
        self.trunk._external_w = {id(u.conv.weight): u.w_fwd for u in self.units}
        self.trunk._external_sync = self.sync
        self._G_averaged = False

    # ------------------------------------------------------------------ parameter arena
    def _flatten_params(self):
        ps = list(self.model.parameters())
        self.off, n = {}, 0
        for p in ps:
            self.off[id(p)] = n
            n += _ceil(p.numel(), 4)            # 16-byte aligned segments
        self.n_params = n
        self.P = torch.zeros(n, device=self.dev, dtype=torch.float32)
        for p in ps:
            o = self.off[id(p)]
            self.P[o:o + p.numel()].copy_(p.data.reshape(-1).float())
            p.data = self.P[o:o + p.numel()].view(p.shape)
        self._params, self._param_ver = ps, sum(p._version for p in ps)
        self.G = torch.zeros_like(self.P)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)

    bn_apply_on_load = os.environ.get("MHE_BN_BWD_ON_LOAD", "1") == "1"
    bn_on_load_max_cin = int(os.environ.get("MHE_BN_BWD_ON_LOAD_MAXC", "128"))   # layer1 / layer2: wider layers lose more on the 128-row tile than the pass costs (measured)

    def _gradient_buckets(self):
        """flat-buffer ranges in the order the reverse pass completes them: [l1, l2, flow, det head] (done before the trunk's
        reverse pass starts), layer4, layer3, [stem, layer1, layer2].  Each range is un-packed and handed to RCCL as soon as
        it is complete, so its all-reduce runs under the rest of the reverse pass."""
        t = self.trunk
        first = lambda mod: self.off[id(next(mod.parameters()))]
        cuts = [0, first(t.layer3), first(t.layer4), self.off[id(self.model.feat_extractor.l1[0].weight)], self.n_params]
        return [(cuts[i], cuts[i + 1]) for i in range(4)]           # index 3 = ready first ... index 0 = ready last

    def _grad_ready(self, i):
        lo, hi = self._bucket_bounds[i]
        self._wgrad_flush()              # the bucket's queued weight gradients, one multi-problem launch per tile shape
        ops.gather(self.raw, self._unpack_idx[lo:hi], self.G[lo:hi])
        if self.comm:
            cap = getattr(self, "_capture", None)
            if cap is not None:              # GraphedStep: the graph ends here, the collective is issued between two graph launches
                cap.cut(("allreduce", i))
            else:
                self._all_reduce_bucket(i)

    def _all_reduce_bucket(self, i):
        """hand bucket i of the flat gradient to the communicator (dist.GradExchange: event-scoped, on its own stream; f32 all-reduce or the
        bf16 all-to-all + all-gather exchange with f32 accumulation)"""
        lo, hi = self._bucket_bounds[i]
        if self._xchg is None:
            from .dist import GradExchange
            self._xchg = GradExchange(self.dist, device=self.dev)
        self._xchg.start(self.G[lo:hi], key=i)

    @property
    def _works(self):           # (tests: how many bucket exchanges are in flight)
        return [None] * (self._xchg.in_flight() if self._xchg is not None else 0)

    def finish_allreduce(self):
        """wait for the gradient buckets' exchanges (sum over ranks; the mean is taken by grad_scale = 1/world)"""
        cap = getattr(self, "_capture", None)
        if cap is not None and self.comm:
            cap.cut(("wait",))
            return
        if self._xchg is not None:
            self._xchg.finish()

    def _pidx(self, p):
        """int64 index tensor shaped like p holding each element's position in the flat buffer"""
        o = self.off[id(p)]
        return torch.arange(o, o + p.numel(), dtype=torch.int64).view(p.shape)

    def grad_of(self, p):
        o = self.off[id(p)]
        return self.G[o:o + p.numel()].view(p.shape)

    def _derived(self, idx, dtype, idx2=None, fallback=False):
        """a tensor refreshed every step as P[idx] (+ P[idx2]); idx int64 with -1 = 0.  fallback=True: a layout only the flow's fallback
        paths read - refreshed when one of them runs (`_need_fallback`)"""
        a = (self._arena_fb if fallback else self._arena)[dtype]
        n = idx.numel()
        room = _ceil(n, 8)                                    # 16-byte aligned views
        if a["buf"] is None:
            a["buf"] = torch.zeros(2 * self.n_params, device=self.dev, dtype=dtype)
        if a["used"] + room > a["buf"].numel():
            raise RuntimeError("TrainStep: derived-operand arena exhausted")
        dst = a["buf"][a["used"]:a["used"] + n].view(idx.shape)
        pad = torch.full((room - n,), -1, dtype=torch.int64)
        a["idx"] += [idx.reshape(-1), pad]
        a["idx2"] += [torch.full((n,), -1, dtype=torch.int64) if idx2 is None else idx2.reshape(-1), pad]
        a["used"] += room
        return dst

    def _raw_slot(self, shape):
        """reserve a raw-gradient slot; returns (offset, getter) - the view exists once the arena is allocated"""
        n = int(np.prod(shape))
        o = self._raw_n
        self._raw_n += _ceil(n, 4)
        return o

    def _raw(self, o, shape):
        return self.raw[o:o + int(np.prod(shape))].view(shape)

    def _map_grad(self, p, raw_index):
        """raw_index: int64 tensor shaped like p giving the raw-arena position of each element's gradient"""
        o = self.off[id(p)]
        self._unpack[o:o + p.numel()] = raw_index.reshape(-1)

    # ------------------------------------------------------------------ trunk tables
    def _build_trunk(self):
        t, T = self.trunk, self.T
        bke = 32 if T == torch.float32 else 64
        self._raw_views = []
        self.units = []

        def add(conv, bn, k, stride, pad, stem=False):
            u = _Unit(conv, bn, k, stride, pad)
            w = conv.weight
            Cout, Cin, KH, KW = w.shape
            idx = self._pidx(w)
            if stem:
                wp = torch.full((64, 8, 24), -1, dtype=torch.int64)
                wp[:, :7, :21] = idx.permute(0, 2, 3, 1).reshape(64, 7, 21)
                u.w_fwd = self._derived(wp.reshape(64, 192), T)
                u.cin_w = 4 if T == torch.float32 else 8              # channel padding of the NHWC image copy
                # bf16: the weight gradient reads the image as PIXEL PAIRS (two neighbours x 3 channels padded to 4 = one 8-channel pixel):
                # a 7 x 4 / stride (2, 1) / pad (3, 2) convolution with 224 weight columns instead of 7 x 7 x 8 = 392 of which 245 multiply
                # zeros (ops.conv_wgrad_rect); raw gradient [64][7][4][2 x 4], column kw = 2 kw' + parity - 1
                u.pairs = T != torch.float32 and os.environ.get("MHE_STEM_WGRAD_PAIRS", "1") == "1"
            else:
                kk = KH * KW * Cin
                f = torch.full((Cout, _ceil(kk, bke)), -1, dtype=torch.int64)
                f[:, :kk] = idx.permute(0, 2, 3, 1).reshape(Cout, kk)
                u.w_fwd = self._derived(f, T)
                # data-gradient operand: W'[ci][kh'][kw'][co] = W[co][ci][KH-1-kh'][KW-1-kw']
                kd = KH * KW * Cout
                d = torch.full((Cin, _ceil(kd, bke)), -1, dtype=torch.int64)
                d[:, :kd] = dgrad_operand_index(idx)
                u.w_dg = self._derived(d, T)
                # the 3x3 / stride-1 units also in the layout of the resident-tile kernel (csrc/conv_halo.hip), forward and data gradient
                # (layer2 / layer3: layer4's 8 x 8 maps are not taken by it, and every table here is gathered every step)
                u.w_halo = u.w_dg_halo = None
                if (KH == 3 and stride == 1 and pad == 1 and T == torch.bfloat16 and Cin % 128 == 0 and Cout % 128 == 0 and Cin <= 256 and Cout <= 256
                        and os.environ.get("MHE_CONV_HALO", "1") == "1"):
                    u.w_halo = self._derived(halo_operand_index(f[:, :kk]), T)
                    u.w_dg_halo = self._derived(halo_operand_index(d[:, :kd]), T)
                u.w_s2 = None
                if KH == 3 and stride == 2 and pad == 1:
                    u.w_s2 = []
                    for tbl in dgrad_s2_operand_indices(idx):
                        f2 = torch.full((Cin, _ceil(tbl.shape[1], bke)), -1, dtype=torch.int64)
                        f2[:, :tbl.shape[1]] = tbl
                        u.w_s2.append(self._derived(f2, T))
                u.cin_w = Cin
            # raw weight gradient [Cout][KH*KW*cin_w]
            if stem and u.pairs:
                u.raw_w = self._raw_slot((Cout, 7 * 4 * 8))
                co, ci, kh, kw = torch.meshgrid(torch.arange(Cout), torch.arange(Cin), torch.arange(7), torch.arange(7), indexing="ij")
                self._map_grad(w, ((co * 7 + kh) * 4 + (kw + 1) // 2) * 8 + ((kw + 1) % 2) * 4 + ci + u.raw_w)
                wshape = (Cout, 7 * 4 * 8)
            else:
                u.pairs = False
                u.raw_w = self._raw_slot((Cout, KH * KW * u.cin_w))
                r = torch.arange(Cout * KH * KW * u.cin_w, dtype=torch.int64).view(Cout, KH, KW, u.cin_w)[..., :Cin] + u.raw_w
                self._map_grad(w, r.permute(0, 3, 1, 2))
                wshape = (Cout, KH * KW * u.cin_w)
            u.raw_g = self._raw_slot((Cout,)); u.raw_b = self._raw_slot((Cout,))
            self._map_grad(bn.weight, torch.arange(Cout) + u.raw_g)
            self._map_grad(bn.bias, torch.arange(Cout) + u.raw_b)

            def views(u=u, shape=wshape):
                u.dw = self._raw(u.raw_w, shape); u.dgamma = self._raw(u.raw_g, (u.cout,)); u.dbeta = self._raw(u.raw_b, (u.cout,))
            self._raw_views.append(views)
            self.units.append(u)
            return u

        self.stem = add(t.conv1, t.bn1, 7, 2, 3, stem=True)
        self.blocks = []
        for li in range(4):
            for blk in getattr(t, f"layer{li + 1}"):
                b = {"kind": blk.kind, "stride": blk.stride, "layer": li + 1}
                if blk.kind == "bottleneck":
                    b["u"] = [add(blk.conv1, blk.bn1, 1, 1, 0), add(blk.conv2, blk.bn2, 3, blk.stride, 1), add(blk.conv3, blk.bn3, 1, 1, 0)]
                else:
                    b["u"] = [add(blk.conv1, blk.bn1, 3, blk.stride, 1), add(blk.conv2, blk.bn2, 3, 1, 1)]
                b["ud"] = add(blk.downsample[0], blk.downsample[1], 1, blk.stride, 0) if blk.downsample is not None else None
                self.blocks.append(b)

    # ------------------------------------------------------------------ dense heads
    def _dense(self, lin, n_pad=None, k_pad=None, want_T=True):
        """tables of one nn.Linear: optional padded copy, transposed (padded) copy for the data gradient,
        raw slots for dW [Npad, Kpad] and db [Npad]"""
        N, K = lin.weight.shape
        Np, Kp = n_pad or N, k_pad or K
        wi = torch.full((Np, Kp), -1, dtype=torch.int64)
        wi[:N, :K] = self._pidx(lin.weight)
        d = {"N": N, "K": K, "Np": Np, "Kp": Kp}
        d["w"] = lin.weight.data if (Np, Kp) == (N, K) else self._derived(wi, torch.float32)
        bi = torch.full((Np,), -1, dtype=torch.int64)
        bi[:N] = self._pidx(lin.bias)
        d["b"] = lin.bias.data if Np == N else self._derived(bi, torch.float32)
        if want_T:
            d["wT"] = self._derived(wi.t().contiguous(), torch.float32)
        d["raw_w"], d["raw_b"] = self._raw_slot((Np, Kp)), self._raw_slot((Np,))
        self._map_grad(lin.weight, (torch.arange(Np * Kp, dtype=torch.int64).view(Np, Kp) + d["raw_w"])[:N, :K])
        self._map_grad(lin.bias, torch.arange(N, dtype=torch.int64) + d["raw_b"])

        def views(d=d):
            d["dw"] = self._raw(d["raw_w"], (d["Np"], d["Kp"])); d["db"] = self._raw(d["raw_b"], (d["Np"],))
        self._raw_views.append(views)
        return d

    def _build_heads(self):
        m = self.model
        self.l1 = self._dense(m.feat_extractor.l1[0])
        self.d0 = self._dense(m.det_head[0])
        self.d2 = self._dense(m.det_head[2], n_pad=32)            # 16 outputs padded to the GEMM's K granule for the data gradient
        # feat_extractor.l2 is dead for MHEnt (hand/network.py:779): its gradient stays zero (-1 in the unpack table)

    # ------------------------------------------------------------------ flow tables
    def _build_flow(self):
        fl = self.model.q_z_giv_i
        from .flows import RealNVP
        self.glow = None
        if not isinstance(fl, RealNVP):
            from .glow import ConditionalGlow
            from .train_glow import GlowPart
            if not isinstance(fl, ConditionalGlow):
                raise NotImplementedError(f"TrainStep: no reverse pass for {type(fl).__name__}")
            self.flow, self.flow_bf16, self.fnets = fl, False, []
            self.glow = GlowPart(self, fl)         # parity unpinned (third-party class absent); eager only: its 45x45
            return                                 # re-parameterisation gradients are computed on the host
        self.flow = fl
        dim, h, ncoup = fl.dim, fl.hidden, len(fl.mask)
        bf16 = fl.compute_dtype == torch.bfloat16 and h % 128 == 0
        self.flow_bf16 = bf16
        # the one-launch forward / reverse kernels (hidden 512) read fragment-major layouts of their own: everything else is a fallback layout
        self.flow_fused_tables = bool(bf16 and h == 512 and self._fb_lazy)
        loc = torch.from_numpy(flow_stream_table(fl.dim, fl.hidden, bf16))
        n0, n1 = h * dim, h * h
        streams, b2, wc, bc1, bc2, nets = [], [], [], [], [], []
        self.fnets = []
        for i in range(ncoup):
            for net in (fl.s[i], fl.t[i]):
                o0, o1, o2 = (self.off[id(net.l[j].weight)] for j in range(3))
                g = torch.where(loc < 0, loc, torch.where(loc < n0, loc + o0, torch.where(loc < n0 + n1, loc - n0 + o1, loc - n0 - n1 + o2)))
                streams.append(g)
                bi = torch.full((64 if bf16 else dim,), -1, dtype=torch.int64)
                bi[:dim] = self._pidx(net.l[2].bias)
                b2.append(bi)
                for j in range(2):
                    wc.append(self._pidx(net.c[j].weight))
                    bc1.append(self._pidx(net.c[j].bias)); bc2.append(self._pidx(net.l[j].bias))
                d = {"net": net}
                # reverse-pass operands (f32): padded W0 [h,64], W1 [h,h] (the parameter itself), padded W2 [64,h] + transposes
                w0i = torch.full((h, 64), -1, dtype=torch.int64); w0i[:, :dim] = self._pidx(net.l[0].weight)
                w2i = torch.full((64, h), -1, dtype=torch.int64); w2i[:dim] = self._pidx(net.l[2].weight)
                b2i = torch.full((64,), -1, dtype=torch.int64); b2i[:dim] = self._pidx(net.l[2].bias)
                fb = self.flow_fused_tables        # (the f32 / plain bf16 layouts below are the fallback paths' when the fragment-major ones exist)
                d["w0"], d["w0T"] = self._derived(w0i, torch.float32, fallback=fb), self._derived(w0i.t().contiguous(), torch.float32, fallback=fb)
                d["w1"], d["w1T"] = net.l[1].weight.data, self._derived(self._pidx(net.l[1].weight).t().contiguous(), torch.float32, fallback=fb)
                if bf16:        # bf16 operand copies for the products that run on bf16 MFMA (all but the two 64-wide f32 ones)
                    d["w1b"] = self._derived(self._pidx(net.l[1].weight), torch.bfloat16, fallback=fb)
                    d["w1Tb"] = self._derived(self._pidx(net.l[1].weight).t().contiguous(), torch.bfloat16, fallback=fb)
                    d["w0b"] = self._derived(w0i, torch.bfloat16, fallback=fb)                           # [h, 64]: XP W0^T
                    d["w2Tb"] = self._derived(w2i.t().contiguous(), torch.bfloat16, fallback=fb)         # [h, 64]: GO W2
                    d["w2b"] = self._derived(w2i, torch.bfloat16, fallback=fb)                           # [64, h]: H1 W2^T (f32 result)
                    d["w0Tb"] = self._derived(w0i.t().contiguous(), torch.bfloat16, fallback=fb)         # [64, h]: G1 W0 (f32 result)
                    # the three [out][k] operands again in MFMA fragment order: what the one-launch reverse chain reads (csrc/flow_rev.hip)
                    d["w1Fb"] = self._derived(ops.mfma_fragment_major(self._pidx(net.l[1].weight).t()), torch.bfloat16)
                    d["w2Fb"] = self._derived(ops.mfma_fragment_major(w2i.t()), torch.bfloat16)
                    d["w0Fb"] = self._derived(ops.mfma_fragment_major(w0i.t()), torch.bfloat16)
                    # ... and the forward's own operands W1 [out][in], W0 (padded) [h][64], W2 (padded) [64][h] (csrc/flow_fwd.hip)
                    d["f1F"] = self._derived(ops.mfma_fragment_major(self._pidx(net.l[1].weight)), torch.bfloat16)
                    d["f0F"] = self._derived(ops.mfma_fragment_major(w0i), torch.bfloat16)
                    d["f2F"] = self._derived(ops.mfma_fragment_major(w2i), torch.bfloat16)
                d["w2"], d["w2T"] = self._derived(w2i, torch.float32, fallback=fb), self._derived(w2i.t().contiguous(), torch.float32, fallback=fb)
                d["b2"] = self._derived(b2i, torch.float32, fallback=fb)
                d["r0"], d["r1"], d["r2"], d["rb2"] = (self._raw_slot(s) for s in ((h, 64), (h, h), (64, h), (64,)))
                self._map_grad(net.l[0].weight, (torch.arange(h * 64, dtype=torch.int64).view(h, 64) + d["r0"])[:, :dim])
                self._map_grad(net.l[1].weight, torch.arange(h * h, dtype=torch.int64).view(h, h) + d["r1"])
                self._map_grad(net.l[2].weight, (torch.arange(64 * h, dtype=torch.int64).view(64, h) + d["r2"])[:dim])
                self._map_grad(net.l[2].bias, torch.arange(dim, dtype=torch.int64) + d["rb2"])
                self.fnets.append(d)
        self.f_stream = self._derived(torch.cat(streams), torch.bfloat16 if bf16 else torch.float32, fallback=self.flow_fused_tables)
        self.f_b2 = self._derived(torch.stack(b2), torch.float32)
        # bf16 mode with the conditioning products in bf16 (forward table, dWc, g_feat): the two f32 copies (2 x 12.6 M elements at C2) would
        # only be gathered every step to be read by nobody
        self.cond_f32 = not (bf16 and fl.tsfm_on % 64 == 0 and self.cond_bwd_bf16)
        self.f_wc = self._derived(torch.cat(wc), torch.float32) if self.cond_f32 else None       # [2*ncoup*2*h, 512]
        self.f_wcb = self._derived(torch.cat(wc), torch.bfloat16) if bf16 and fl.tsfm_on % 64 == 0 else None      # forward operand in the bf16 mode
        self.f_bc = self._derived(torch.cat(bc1), torch.float32, torch.cat(bc2))       # c_j.bias + l_j.bias
        self.f_wcT = self._derived(torch.cat(wc).t().contiguous(), torch.float32) if self.cond_f32 else None     # [512, slots*h]
        slots = 4 * ncoup
        self.f_slots = slots
        raw_wc, raw_bc = self._raw_slot((slots * h, fl.tsfm_on)), self._raw_slot((slots * h,))
        k = 0
        for i in range(ncoup):
            for net in (fl.s[i], fl.t[i]):
                for j in range(2):
                    self._map_grad(net.c[j].weight, torch.arange(h * fl.tsfm_on, dtype=torch.int64).view(h, fl.tsfm_on) + raw_wc + k * h * fl.tsfm_on)
                    bidx = torch.arange(h, dtype=torch.int64) + raw_bc + k * h
                    self._map_grad(net.c[j].bias, bidx); self._map_grad(net.l[j].bias, bidx)
                    k += 1

        def views():
            self.dwc = self._raw(raw_wc, (slots * h, fl.tsfm_on)); self.dbc = self._raw(raw_bc, (slots * h,))
            for d in self.fnets:
                d["dw0"], d["dw1"], d["dw2"], d["db2"] = (self._raw(d[k_], s) for k_, s in (("r0", (h, 64)), ("r1", (h, h)), ("r2", (64, h)), ("rb2", (64,))))
        self._raw_views.append(views)

    # ------------------------------------------------------------------ per-step plumbing
    def _gather(self, a):
        if a.get("aff") is not None:
            bs, msk, segs = a["aff"]
            for kind, lo, hi in segs:
                if kind == "aff":
                    ops.gather_affine8(self.P, bs[lo:hi], msk[lo:hi], a["view"][8 * lo:8 * hi])
                else:
                    ops.gather(self.P, a["idx"][8 * lo:8 * hi], a["view"][8 * lo:8 * hi])
        else:
            ops.gather(self.P, a["idx"], a["view"], a["idx2"])

    def repack(self):
        for a in self._arena.values():
            if a["idx"].numel():
                self._gather(a)
        if self._fb_keep:
            self._repack_fallback()
        else:
            self._fb_stale = True
            self._poison_fallback()

    def _poison_fallback(self):
        """debug mode (MHE_POISON_STALE_TABLES=1, tests): a fallback layout left behind by a repack is filled with NaN, so a reader that
        did not go through _need_fallback() fails loudly instead of computing with an earlier step's weights"""
        if self._poison_stale:
            for a in self._arena_fb.values():
                if a["idx"].numel():
                    a["view"].fill_(float("nan"))

    def _repack_fallback(self):
        for a in self._arena_fb.values():
            if a["idx"].numel():
                self._gather(a)
        self._fb_stale = False

    def _need_fallback(self):
        """called by every path that reads a fallback layout (this class's coupling-by-coupling passes, the modules' own forward / sample
        paths through sync_all): brings the layouts up to the current parameters if the last repack skipped them, and keeps them fresh from
        now on (this process evidently runs such steps)"""
        if self._fb_stale:
            self._repack_fallback()
        self._fb_keep = True

    def sync_all(self):
        """sync() for the modules' own paths, which may read any layout (RealNVP._packed: the second-generation kernel's stream)"""
        self.sync()
        self._need_fallback()

    def sync(self):
        """refresh every derived operand layout if someone else (torch.optim through the attach() bridge, load_state_dict)
        wrote the parameters since the last repack; called by every entry point that reads the packs - this class's
        forward() and the modules' own forward / sample / log_prob paths (ResNetTrunk.forward, RealNVP._packed)"""
        ver = sum(p._version for p in self._params)
        if ver != self._param_ver:
            self.repack()
            if self.glow is not None:
                self.glow.invalidate()
            self._param_ver = ver

    def _buf(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        b = self._ws.get(key)
        if b is None:
            b = self._ws[key] = torch.empty(shape, device=self.dev, dtype=dtype)
        return b

    # ------------------------------------------------------------------ trunk forward / backward
    def _unit_fwd(self, u, x, pool):
        st = pool.take(u.cout)
        y = ops.conv2d_nhwc(x, u.w_fwd, u.k, u.k, u.stride, u.pad, stats=st)
        return self._bn_tape(u, x, y, st)

    def _bn_tape(self, u, x, y, st):
        count = y.numel() // u.cout
        bn = u.bn
        u.scale, u.shift, u.mi = ops.bn_finalize(st, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, count, BN_MOMENTUM,
                                                 BN_EPS, want_mean_invstd=True, clear=True, num_batches_tracked=bn.num_batches_tracked)
        u.x, u.y = x, y
        return y

    def _trunk_forward(self, x):
        T = self.T
        pool = self.trunk._stats_pool(self.dev)       # self-cleaning arena shared with the module's own forward
        u = self.stem
        if u.pairs:
            if x.shape[3] % 2 or x.shape[2] % 2:
                raise ValueError("TrainStep (bf16): the stem's weight gradient reads pixel pairs - even image sizes only (MHE_STEM_WGRAD_PAIRS=0 lifts this)")
            self.x_nhwc = ops.nchw_to_nhwc(x, T, cpad=4).view(x.shape[0], x.shape[2], x.shape[3] // 2, 8)
        else:
            self.x_nhwc = ops.nchw_to_nhwc(x, T)
        st = pool.take(64)
        y0 = ops.stem_conv7x7s2(x, u.w_fwd, T, stats=st)
        self._bn_tape(u, self.x_nhwc, y0, st)
        if self.stem_pool_fused:
            # pool straight from the raw stem output, BatchNorm + ReLU on the load: the normalised full-resolution copy is never written
            if self.stem_pooled_sums:       # ... and keep the raw winners: the reverse pass takes the BatchNorm-reverse sums from the pooled tensors
                a, self.pool_idx, self.pool_win = ops.maxpool3x3s2_idx_win(y0, u.scale, u.shift)
                self.pool_out = a
            else:
                a, self.pool_idx = ops.maxpool3x3s2_idx(y0, u.scale, u.shift)
            self.r0 = None
        else:
            self.r0 = ops.bn_act(y0, u.scale, u.shift, relu=True)
            a, self.pool_idx = ops.maxpool3x3s2_idx(self.r0)
        pending = None          # (raw conv3 output, its unit, identity tensor, downsample unit | None): a block tail not yet evaluated
        raw_prev = None         # (conv1's raw output, its unit): normalisation left to conv2's operand load
        fuse = self.trunk.fuse_tail
        for bi, b in enumerate(self.blocks):
            us = b["u"]
            if pending is not None and isinstance(pending[0], str):
                # ... with the previous block's conv3 evaluated again inside the same kernel: its raw output was never written (the reverse
                # pass evaluates it once more when it gets there, _ensure_y)
                _, y2_p, bn2_p, ul_p, idt_p, ud_p = pending
                u0 = us[0]
                st = pool.take(u0.cout)
                # (+ [a > 0] as bits: the reverse pass's gate at a sixteenth of a's bytes, MHE_GATE_BITS=0: it reads a)
                a, y, abits = ops.bottleneck_tail(y2_p, bn2_p, ul_p.w_fwd, (ul_p.scale, ul_p.shift), idt_p,
                                                  None if ud_p is None else (ud_p.scale, ud_p.shift), u0.w_fwd, stats=st, want_bits=True)
                b["a_bits"] = abits if self.gate_bits else None
                self._bn_tape(u0, a, y, st)
                self.blocks[bi - 1]["out"] = a
                pending = None
                b["a"] = a
                if self._halo_ok(us[1], y):      # conv2 normalises its operand on its own load and writes it out on the way
                    h, raw_prev = None, (y, u0)
                    b["acts"] = []
                else:
                    h = ops.bn_act(y, u0.scale, u0.shift, relu=True)
                    b["acts"] = [h]
                rest = us[1:-1]
            elif pending is not None:
                # relu(bn3(y3) + identity) of the previous block is evaluated inside this conv1's operand load, which also
                # writes it out once (this block's input / identity and the reverse pass's ReLU mask): one read of the widest
                # tensor of the block saved, as in the inference path (resnet.py)
                yl_p, ul_p, idt_p, ud_p = pending
                a = torch.empty_like(yl_p)
                u0 = us[0]
                st = pool.take(u0.cout)
                y = ops.conv1x1_residual_in(yl_p, idt_p, u0.w_fwd, ul_p.scale, ul_p.shift, None if ud_p is None else ud_p.scale,
                                            None if ud_p is None else ud_p.shift, a_out=a, stats=st)
                self._bn_tape(u0, a, y, st)
                self.blocks[bi - 1]["out"] = a
                pending = None
                b["a"], b["a_bits"] = a, None
                if self._halo_ok(us[1], y):      # conv2 normalises its operand on its own load and writes it out on the way
                    h, raw_prev = None, (y, u0)
                    b["acts"] = []
                else:
                    h = ops.bn_act(y, u0.scale, u0.shift, relu=True)
                    b["acts"] = [h]
                rest = us[1:-1]
            else:
                b["a"], b["a_bits"] = a, None
                h = a
                b["acts"] = []
                rest = us[:-1]
            ul = us[-1]
            nxt = self.blocks[bi + 1] if bi + 1 < len(self.blocks) else None
            h_by_gram = False
            for u in rest:
                if raw_prev is not None:
                    y1, up = raw_prev
                    raw_prev = None
                    h = torch.empty_like(y1)
                    st = pool.take(u.cout)
                    y = ops.conv3x3_halo(y1, u.w_halo, up.scale, up.shift, relu_in=True, a_out=h, stats=st)
                    self._bn_tape(u, h, y, st)
                    b["acts"].append(h)
                else:
                    y = self._unit_fwd(u, h, pool)
                # conv2's output of a block whose conv3 runs on Gram statistics: the Gram launch below reads it raw anyway and writes the
                # normalised tensor on the way (no bn_act pass of its own)
                h_by_gram = (u is us[-2] and self._foldable(b, nxt, us, y) and (self.conv3_fold and self.fuse_bn_reduce or
                             ops.bottleneck_tail_supported(y.shape[0], y.shape[1], y.shape[2], y.shape[3], nxt["u"][0].cout)))
                h = torch.empty_like(y) if h_by_gram else ops.bn_act(y, u.scale, u.shift, relu=True)
                b["acts"].append(h)
            # layer1 / layer2 bottlenecks (bf16, 64 / 128 bottleneck channels): conv3 is not run here at all - bn3's batch statistics come
            # from the Gram matrix of its input, the tail kernel of the next block evaluates it on the fly (as the module's own forward
            # does, resnet.py), and the reverse pass evaluates it once when it needs it: a write + a read of the block's widest tensor
            # traded for one plain 1x1 launch in the reverse pass (MHE_TRAIN_RECOMPUTE=0: conv3 written in the forward pass)
            foldable = self._foldable(b, nxt, us, h)
            recompute = foldable and ops.bottleneck_tail_supported(h.shape[0], h.shape[1], h.shape[2], h.shape[3], nxt["u"][0].cout)
            # (a block whose tail the fused kernel cannot take - layer2's last, which feeds layer3 - still runs conv3 for the forward's sake,
            # but on the Gram statistics as well, so that the REVERSE pass can do without y3: it is dropped from the tape)
            semi = foldable and not recompute and self.conv3_fold and self.fuse_bn_reduce
            if recompute or semi:
                u2, bn3 = us[-2], ul.bn
                gbufs = pool.gram(ul.cin)
                if self.conv3_fold:      # the Gram totals of THIS block stay for the reverse pass (csrc/conv_fold.hip)
                    tot = self._ws.get(("gram_tot", bi))
                    if tot is None:
                        tot = self._ws[("gram_tot", bi)] = ops.gram_workspace(ul.cin, self.dev)
                    gbufs, ul.gram_tot = (gbufs[0], tot), tot
                else:
                    ul.gram_tot = None
                ul.scale, ul.shift, ul.mi = ops.conv1x1_gram_bn(u2.y, u2.scale, u2.shift, ul.w_fwd, bn3.weight.data, bn3.bias.data, bn3.running_mean,
                                                                bn3.running_var, gbufs, BN_MOMENTUM, BN_EPS,
                                                                num_batches_tracked=bn3.num_batches_tracked, want_mean_invstd=True,
                                                                a_out=h if h_by_gram else None)
                ul.x, ul.y = h, None
                yl = ops.conv2d_nhwc(h, ul.w_fwd, 1, 1, 1, 0) if semi else None
            else:
                yl = self._unit_fwd(ul, h, pool)
            ud = b["ud"]
            yd = self._shortcut_fwd(bi, b, ud, pool) if ud is not None else None
            if recompute:
                pending = ("re", us[-2].y, (us[-2].scale, us[-2].shift), ul, yd if ud is not None else b["a"], ud)
                continue
            if fuse and nxt is not None and nxt["kind"] == "bottleneck" and b["kind"] == "bottleneck":
                pending = (yl, ul, yd if ud is not None else b["a"], ud)
                continue
            if ud is not None:
                a = ops.bn_act(yl, ul.scale, ul.shift, yd, ud.scale, ud.shift, relu=True)
            else:
                a = ops.bn_act(yl, ul.scale, ul.shift, b["a"], relu=True)
            b["out"] = a
        pool.done()
        self.a_last = a
        return ops.avgpool(a)

    def _bn_bwd(self, u, g, a, pool, stats=None):
        """g: gradient w.r.t. the unit's BatchNorm OUTPUT (already ReLU-gated by its producer when a is None).  stats: the
        reverse sums already accumulated by the producer's epilogue (then only finalize + apply run here)."""
        return ops.bn_backward(g, a, u.y, u.mi, u.bn.weight.data, stats if stats is not None else pool.take(u.cout), u.dgamma, u.dbeta,
                               reduced=stats is not None)

    def _shortcut_fwd(self, bi, b, ud, pool):
        """the block's shortcut convolution + its BatchNorm's batch statistics.  Layer1's (1x1, stride 1, 64 input channels): the statistics
        from the Gram matrix of the block's input, whose totals stay for the reverse pass - there the BatchNorm reverse then needs neither the
        shortcut's raw output nor a pass of its own over three block-wide tensors (csrc/conv_fold.hip, as for conv3; MHE_SHORTCUT_FOLD=0: as the others)"""
        a = b["a"]
        ud.fold_rev = False
        if not (self.shortcut_fold and self.conv3_fold and self.fuse_bn_reduce and ud.k == 1 and ud.stride == 1 and ud.cin in (64, 128)
                and a.dtype == torch.bfloat16 and (a.numel() // ud.cin) % 128 == 0 and bi + 1 < len(self.blocks) and self.blocks[bi + 1]["ud"] is None
                and self.blocks[bi + 1]["u"][0].k == 1):
            return self._unit_fwd(ud, a, pool)
        tot = self._ws.get(("gram_tot_ds", bi))
        if tot is None:
            tot = self._ws[("gram_tot_ds", bi)] = ops.gram_workspace(ud.cin, self.dev)
            self._ws["ones_c"] = torch.ones(4096, device=self.dev)
        bn = ud.bn
        ud.scale, ud.shift, ud.mi = ops.conv1x1_gram_bn(a, self._ws["ones_c"][:ud.cin], self._zeros_c[:ud.cin], ud.w_fwd, bn.weight.data, bn.bias.data,
                                                        bn.running_mean, bn.running_var, (pool.gram(ud.cin)[0], tot), BN_MOMENTUM, BN_EPS,
                                                        num_batches_tracked=bn.num_batches_tracked, want_mean_invstd=True)
        ud.gram_tot, ud.fold_rev = tot, True
        ud.x = a
        ud.y = ops.conv2d_nhwc(a, ud.w_fwd, 1, 1, 1, 0)
        return ud.y

    def _halo_ok(self, u, x):
        return (self.conv_halo and getattr(u, "w_halo", None) is not None and u.k == 3 and u.stride == 1 and x.dtype == torch.bfloat16
                and ops.conv3x3_halo_supported(x.shape[0], x.shape[1], x.shape[2], u.cin, u.cout))

    def _foldable(self, b, nxt, us, h):
        """a bottleneck of layer1 / layer2 whose conv3 + bn3 can run on the Gram statistics of conv3's input (h: that input, or conv2's raw
        output - same shape)"""
        ul = us[-1]
        return (self.train_recompute and self.trunk.fuse_tail and nxt is not None and nxt["kind"] == "bottleneck" and b["kind"] == "bottleneck"
                and len(us) == 3 and ul.k == 1 and ul.stride == 1 and h.dtype == torch.bfloat16 and us[-2].y is not None
                and nxt["u"][0].k == 1 and nxt["u"][0].stride == 1 and ul.cin in (64, 128) and (h.numel() // ul.cin) % 128 == 0)

    def _ensure_y(self, u):
        """the raw output of a unit whose forward launch was skipped (conv3 of a layer1 / layer2 bottleneck): evaluated now, bit-identical
        to what the fused tail kernel worked with"""
        if u.y is None:
            u.y = ops.conv2d_nhwc(u.x, u.w_fwd, u.k, u.k, u.stride, u.pad)
            u.y_recomputed = True
        return u.y

    def _wgrad(self, u, gy):
        """dW of a trunk unit.  bf16 trunk: QUEUED - a gradient bucket's weight gradients do not depend on one another, so they are launched
        together when the bucket completes (ops.conv_wgrad_multi: the layers share the chip, every pixel range is cut 4 - 8 ways instead of
        28 - 64, layer4 not at all - the partial-slab traffic of one launch per layer, 5.4 GB of the step's 98 GB, falls accordingly).  The
        queue holds x and gy alive; nothing in the reverse pass writes into either after this point (gy is this unit's own tensor, x a
        forward activation)."""
        if self.wgrad_multi and gy.dtype == torch.bfloat16:
            self._wq.append((u.x, gy, u.k, u.k, u.stride, u.pad, u.dw))
        else:
            ops.conv_wgrad(u.x, gy, u.k, u.k, u.stride, u.pad, u.dw)

    def _wgrad_flush(self):
        if self._wq:
            ops.conv_wgrad_multi(self._wq)
            self._wq = []

    def _dgrad(self, u, gy, residual=None, gate=True, consumers=(), pool=None, res_half=False, coarse=False, mask_bits=None):
        """gradient w.r.t. the pre-activation of the unit's input (+ residual): every unit input in the trunk is a post-ReLU
        tensor, so the ReLU gate [x > 0] is applied in the producing kernel's epilogue and the BatchNorm reverse passes
        downstream read one tensor less"""
        bn = None
        if gate and consumers and self.fuse_bn_reduce:
            # the BatchNorm units that consume this gradient: their reverse sums are accumulated by this kernel's epilogue
            # (a unit whose raw output was never written and whose reverse runs on the Gram statistics, csrc/conv_fold.hip, needs sum g only:
            # the gate tensor - same shape, read by this epilogue anyway - stands in for its output; the second sum is not used)
            dummy = [c.y is None or getattr(c, "fold_rev", False) for c in consumers]
            bn = [(u.x if dm else c.y, c.mi, pool.take(c.cout)) for c, dm in zip(consumers, dummy)]
            for c, (_, _, st), dm in zip(consumers, bn, dummy):
                c.rev_stats = st
                c.rev_dummy = dm
        if (self.conv_halo_dg and gate and residual is None and mask_bits is None and len(consumers) <= 1 and getattr(u, "w_dg_halo", None) is not None
                and u.k == 3 and u.stride == 1 and gy.dtype == torch.bfloat16
                and ops.conv3x3_halo_supported(gy.shape[0], gy.shape[1], gy.shape[2], u.cout, u.cin)):
            return ops.conv3x3_halo(gy, u.w_dg_halo, mask=u.x, bn=None if bn is None else bn[0])
        return conv_dgrad(gy, u.w_dg, u.k, u.stride, u.pad, u.x.shape[1], u.x.shape[2], residual, u.x if gate else None, bn,
                          w_s2=getattr(u, "w_s2", None), res_half=res_half, coarse=coarse, mask_bits=mask_bits)

    def _trunk_backward(self, g_f):
        # the reverse pass's BatchNorm sums: ONE arena kept across steps; a pass zeroes the slice the previous pass used (ResNet-50 takes
        # ~27k channels = 55 MB of fixed-point words; round 4 allocated and zeroed a fresh 134 MB arena every step - and every graph replay)
        pool = getattr(self, "_rev_pool", None)
        if pool is None:
            pool = self._rev_pool = resnet._StatsPool(self.dev, channels=65536, persistent=True)
            pool.high = 0
        else:
            pool.buf[:pool.high].zero_()
        pool.off = 0
        self.n_fold = 0                 # blocks whose conv3 + bn3 were reversed on the Gram statistics in this pass
        self.n_fold_ds = 0              # ... and shortcuts
        B, Hh, Ww, Cc = self.a_last.shape
        # g is always the gradient w.r.t. the block output's PRE-ReLU value: the gate is applied where g is produced
        g = ops.avgpool_bwd(g_f, Hh * Ww, self.T, mask=self.a_last.view(B, Hh * Ww, Cc)).view(B, Hh, Ww, Cc)
        for bi in range(len(self.blocks) - 1, -1, -1):
            b = self.blocks[bi]
            if bi + 1 < len(self.blocks) and self.blocks[bi + 1]["layer"] != b["layer"] and self.blocks[bi + 1]["layer"] >= 3:
                self._grad_ready(self.blocks[bi + 1]["layer"] - 2)      # layer4 complete -> bucket 2, layer3 -> bucket 1
            us, ud = b["u"], b["ud"]
            ul = us[-1]
            # conv3 + bn3 reversed on the forward's Gram statistics (layer1 / layer2, csrc/conv_fold.hip): neither y3 nor gy3 exists
            fold = (ul.y is None and getattr(ul, "gram_tot", None) is not None and getattr(ul, "rev_stats", None) is not None
                    and getattr(ul, "rev_dummy", False) and self.fuse_bn_reduce)
            if not fold:
                self._ensure_y(ul)
            # bottleneck conv3 (1x1, stride 1): its BatchNorm reverse is applied in the operand load of its own data gradient
            # (ops.conv1x1_dgrad_bn_apply) instead of by a pass over three block-wide tensors
            # (the register-staged kernel pays for it up to 128 bottleneck channels; the wide layers take it where the transfer-wave
            # kernel, csrc/conv_tail.hip, runs their data gradient)
            on_load = self.bn_apply_on_load and len(us) == 3 and ul.k == 1 and ul.stride == 1 and (
                ul.cin <= self.bn_on_load_max_cin or (self.bn_on_load_wide and ops.conv_tile_choice(
                    g.shape[0], g.shape[1], g.shape[2], ul.cout, ul.cin, 1, 1, 0, g.dtype, 2) == 10))
            if fold:
                self.n_fold += 1
                Cn, Cb = ul.cout, ul.cin
                D = self._ws.get(("foldD", Cn, Cb))
                if D is None:
                    D = self._ws[("foldD", Cn, Cb)] = torch.zeros(Cn, Cb, device=self.dev)      # cleared by the fold kernel on its way out
                ops.conv_wgrad(ul.x, g, 1, 1, 1, 0, D)
                # [(k2 W)^T | W^T diag(k1) W]: the weights of one data-gradient launch on [g | A] (MHE_CONV3_FOLD_CAT=0: two launches, the
                # Cb x Cb product on A as the residual of the one on g)
                cat = self.conv3_fold_cat
                fw = (self._buf(f"fold_wcat{Cn}", (Cb, Cn + Cb), torch.bfloat16) if cat else self._buf(f"fold_wdg{Cn}", (Cb, Cn), torch.bfloat16),
                      None if cat else self._buf(f"fold_S{Cb}", (Cb, Cb), torch.bfloat16), self._buf(f"fold_c0{Cb}", (Cb,)))
                ops.conv3_bn_fold(D, ul.w_fwd, ul.gram_tot, ul.rev_stats, ul.bn.weight.data, ul.mi, g.numel() // Cn, ul.dgamma, ul.dbeta, ul.dw,
                                  fw[0], fw[1], fw[2], self._buf(f"fold_coef{Cn}", (2 * Cn,)))
                gy = None
            elif on_load:
                rs = getattr(ul, "rev_stats", None)
                coef = ops.bn_backward(g, None, ul.y, ul.mi, ul.bn.weight.data, rs if rs is not None else pool.take(ul.cout), ul.dgamma,
                                       ul.dbeta, reduced=rs is not None, coef_only=True)
                gy = None
            else:
                gy = self._bn_bwd(ul, g, None, pool, stats=getattr(ul, "rev_stats", None))
            ud_fold = (ud is not None and getattr(ud, "fold_rev", False) and getattr(ud, "rev_stats", None) is not None
                       and getattr(ud, "rev_dummy", False))
            if ud_fold:
                # the shortcut's BatchNorm reverse on the Gram statistics of the block's input: D = g^T a by a weight-gradient launch on g itself,
                # dW / dgamma / dbeta and the weights [(k2 W)^T | W^T diag(k1) W] of ONE ungated data-gradient launch on [g | a] out of the fold
                Cn, Cb = ud.cout, ud.cin
                D = self._ws.get(("foldD", Cn, Cb))
                if D is None:
                    D = self._ws[("foldD", Cn, Cb)] = torch.zeros(Cn, Cb, device=self.dev)
                ops.conv_wgrad(ud.x, g, 1, 1, 1, 0, D)
                # (buffers of its own: conv3's fold of this block has run already, its data-gradient launch - further down - has not)
                wcat, c0 = self._buf(f"fold_ds_wcat{Cn}", (Cb, Cn + Cb), torch.bfloat16), self._buf(f"fold_ds_c0{Cb}", (Cb,))
                ops.conv3_bn_fold(D, ud.w_fwd, ud.gram_tot, ud.rev_stats, ud.bn.weight.data, ud.mi, g.numel() // Cn, ud.dgamma, ud.dbeta, ud.dw,
                                  wcat, None, c0, self._buf(f"fold_ds_coef{Cn}", (2 * Cn,)))
                self.n_fold_ds += 1
                skip = ops.conv2d_nhwc(g, wcat, 1, 1, 1, 0, xcat=ud.x, out_shift=c0)
                half_skip = False
            elif ud is not None:
                gyd = self._bn_bwd(ud, g, None, pool, stats=getattr(ud, "rev_stats", None))
                self._wgrad(ud, gyd)
                # summed with the main branch before the gate; a stride-2 shortcut's gradient stays on its coarse grid and the main
                # branch's data gradient adds it at the even positions (bottleneck: conv1 is 1x1 stride 1, so that launch takes it)
                half_skip = ud.stride == 2 and ud.k == 1 and us[0].stride == 1 and bi > 0
                skip = self._dgrad(ud, gyd, gate=False, coarse=half_skip)
            else:
                skip = g
            for j in range(len(us) - 1, 0, -1):
                u = us[j]
                if isinstance(gy, tuple):
                    # a 3x3 unit of layer2 / layer3: its BatchNorm reverse applied on the operand load of its own data gradient (the
                    # resident-tile kernel's transfer waves, csrc/conv_halo.hip), gy written on the way for the weight gradient
                    graw, coef = gy
                    cons = us[j - 1]
                    bn = None
                    if self.fuse_bn_reduce:
                        cons.rev_stats = pool.take(cons.cout)
                        bn = (cons.y, cons.mi, cons.rev_stats)
                    gy = torch.empty_like(graw)
                    ga = ops.conv3x3_halo_dgrad_bn(graw, u.y, coef, u.w_dg_halo, u.x, gy_out=gy, bn=bn)
                    self._wgrad(u, gy)
                elif gy is None:                             # conv3 with its BatchNorm reverse on load
                    cons = us[j - 1]
                    bn = None
                    if self.fuse_bn_reduce:
                        cons.rev_stats = pool.take(cons.cout)
                        bn = [(cons.y, cons.mi, cons.rev_stats)]
                    if fold:
                        # gy3 W = g (k2 W) + A (W^T diag(k1) W) + k0^T W: the Cb x Cb product on conv3's input as the residual, the constant
                        # as the bias of ONE data-gradient launch on g; the weight gradient came out of the fold
                        if fw[1] is None:
                            ga = ops.conv2d_nhwc(g, fw[0], 1, 1, 1, 0, mask=u.x, bn=bn, out_shift=fw[2], xcat=u.x)
                        else:
                            t_res = ops.conv2d_nhwc(u.x, fw[1], 1, 1, 1, 0)
                            ga = ops.conv2d_nhwc(g, fw[0], 1, 1, 1, 0, residual=t_res, mask=u.x, bn=bn, out_shift=fw[2])
                    else:
                        gy = torch.empty_like(g)
                        ga = ops.conv1x1_dgrad_bn_apply(g, u.y, coef, u.w_dg, gy, u.x, bn, zeros=self._zeros_c[:u.cout])
                        self._wgrad(u, gy)
                else:
                    self._wgrad(u, gy)
                    ga = self._dgrad(u, gy, consumers=(us[j - 1],), pool=pool)
                nu = us[j - 1]
                if (j - 1 >= 1 and self.conv_halo_dg and self.halo_bn_on_load and getattr(nu, "w_dg_halo", None) is not None and nu.k == 3
                        and nu.stride == 1 and ga.dtype == torch.bfloat16
                        and ops.conv3x3_halo_supported(ga.shape[0], ga.shape[1], ga.shape[2], nu.cout, nu.cin)):
                    rs = getattr(nu, "rev_stats", None)
                    gy = (ga, ops.bn_backward(ga, None, nu.y, nu.mi, nu.bn.weight.data, rs if rs is not None else pool.take(nu.cout), nu.dgamma,
                                              nu.dbeta, reduced=rs is not None, coef_only=True))
                else:
                    gy = self._bn_bwd(nu, ga, None, pool, stats=getattr(nu, "rev_stats", None))
            self._wgrad(us[0], gy)
            first = bi == 0            # the first block's input is the max-pooled stem output (>= 0; the pool's reverse gates it)
            prev = self.blocks[bi - 1] if bi else None
            if prev is not None and not (self.conv3_fold and self.fuse_bn_reduce and getattr(prev["u"][-1], "gram_tot", None) is not None):
                self._ensure_y(prev["u"][-1])
            cons = () if first else tuple(x for x in (prev["u"][-1], prev["ud"]) if x is not None)
            g = self._dgrad(us[0], gy, residual=skip, gate=not first, consumers=cons, pool=pool, res_half=ud is not None and half_skip,
                            mask_bits=b.get("a_bits"))
        for u in self.units:
            u.rev_stats, u.rev_dummy = None, False
            if getattr(u, "y_recomputed", False):
                u.y, u.y_recomputed = None, False
        u = self.stem
        if self.stem_pool_fused:
            # pool scatter + ReLU gate (recomputed from the raw output) + the BatchNorm-reverse sums in one pass
            st = pool.take(u.cout)
            if self.stem_bwd_two_pass:
                # the same walk twice - sums, then the BatchNorm reverse applied where the scattered gradient is formed: that gradient (as
                # large as the stem's output: 0.54 GB at C2) is never written or read back (MHE_STEM_BWD_TWO_PASS=0: one walk + an apply pass)
                if self.stem_pooled_sums:   # the sums from the pooled tensors (the gradient is non-zero at the pool's winners only): no first walk
                    ops.pooled_bn_sums(g, self.pool_out, self.pool_win, u.mi, st)
                else:
                    ops.maxpool3x3s2_bwd_bn(g, self.pool_idx, u.y, u.scale, u.shift, u.mi, st, want_gx=False)
                coef = ops.bn_bwd_coef(st, u.bn.weight.data, u.mi, u.dgamma, u.dbeta, u.y.numel() // u.cout)
                gy0 = ops.maxpool3x3s2_bwd_bn_apply(g, self.pool_idx, u.y, u.scale, u.shift, u.mi, coef)
            else:
                g_r0 = ops.maxpool3x3s2_bwd_bn(g, self.pool_idx, u.y, u.scale, u.shift, u.mi, st)
                gy0 = self._bn_bwd(u, g_r0, None, pool, stats=st)
        else:
            g_r0 = ops.maxpool3x3s2_bwd(g, self.pool_idx, self.r0.shape[1], self.r0.shape[2])
            gy0 = self._bn_bwd(u, g_r0, self.r0, pool)
        if u.pairs:
            ops.conv_wgrad_rect(self.x_nhwc, gy0, 7, 4, 2, 1, 3, 2, u.dw)
        else:
            ops.conv_wgrad(self.x_nhwc, gy0, 7, 7, 2, 3, u.dw)
        self._grad_ready(0)

    # ------------------------------------------------------------------ flow reverse
    def _flow_backward(self, x_out, cond, g_x, g_logp, N, B, N_all=None):
        """N: hypotheses per image among the rows; N_all: hypotheses per image the means are taken over (differs only under
        hypothesis sharding)"""
        N_all = N_all or N
        fl = self.flow
        h, dim, ncoup = fl.hidden, fl.dim, len(fl.mask)
        R = x_out.shape[0]
        cstride = self.f_slots * h
        XP = self._buf("XP", (R, 64))
        Hb = [[self._buf(f"H{n}{j}", (R, h)) for j in range(2)] for n in range(2)]
        O = [self._buf(f"O{n}", (R, 64)) for n in range(2)]
        GO = [self._buf(f"GO{n}", (R, 64)) for n in range(2)]
        GX = [self._buf(f"GX{n}", (R, 64)) for n in range(2)]
        G2, G1 = self._buf("G2", (R, h)), self._buf("G1", (R, h))
        xa, xb = self._buf("xa", (R, dim)), self._buf("xb", (R, dim))
        ga, gb = self._buf("ga", (R, dim)), self._buf("gb", (R, dim))
        gpart = self._buf("gpart", (R, dim))
        Gc = self._buf("Gcond", (B, cstride))                     # gradient of the conditioning table, all nets / layers
        cflat = cond.view(B, cstride)
        x_cur, g_cur = x_out, g_x
        mixed = self.flow_bf16
        if mixed:
            bf = torch.bfloat16
            H1b = [self._buf(f"H1b{n}", (R, h), bf) for n in range(2)]
            P2b, G2b, GH1b = self._buf("P2b", (R, h), bf), self._buf("G2b", (R, h), bf), self._buf("GH1b", (R, h), bf)
            v4 = lambda t: t.view(R, 1, 1, t.shape[1])
        if mixed:
            # bf16 performance mode: every product except the two that feed exp/tanh (H1 W2^T -> s, t) or the flow variable's own
            # gradient chain (G1 W0 -> GX) takes bf16 operands with f32 accumulation - as the forward kernel does; the leaky-ReLU
            # reverse is fused with the per-image sums that give the conditioning table's gradient
            XPb, P0b = self._buf("XPb", (R, 64), bf), self._buf("P0b", (R, h), bf)
            H2b = [self._buf(f"H2b{n}", (R, h), bf) for n in range(2)]
            GOb = [self._buf(f"GOb{n}", (R, 64), bf) for n in range(2)]
            G1b = self._buf("G1b", (R, h), bf)
            GcT = self._buf("GcondT", (cstride, B))              # the same sums as Gc, [column][image]: split-K operand of g_feat
            self._GcT = GcT
            self._Gc_packed = None
            kept = getattr(self, "_flow_kept", None)
            if kept is not None and kept[0].shape[1] != R:
                kept = None
            # grouped weight gradients (ops.conv_wgrad_batched): with the forward's activations kept, every net's reverse operands are kept
            # too (GO, G2, G1, the masked inputs: 0.9 GB at C2) and the 72 per-net weight-gradient launches (4 - 16 output tiles each,
            # 30 - 55 us apiece) become four grouped ones after the chain - MHE_FLOW_WGRAD_GROUPED=0: per net, as the chain goes
            grouped = kept is not None and os.environ.get("MHE_FLOW_WGRAD_GROUPED", "1") == "1"
            if grouped:
                GOb_all = self._buf("GOb_all", (2 * ncoup, R, 64), bf)
                G2b_all, G1b_all = self._buf("G2b_all", (2 * ncoup, R, h), bf), self._buf("G1b_all", (2 * ncoup, R, h), bf)
                XPb_all = self._buf("XPb_all", (ncoup, R, 64), bf)
            # the whole data-gradient chain in one launch (csrc/flow_rev.hip): 64 hypotheses per image, one workgroup per image.
            # MHE_FLOW_REV_FUSED=0: coupling by coupling (13 launches each)
            fused = (grouped and os.environ.get("MHE_FLOW_REV_FUSED", "1") == "1" and N == N_all and R == 64 * B
                     and ops.flow_reverse_chain_supported(R, B, dim, h, ncoup))
            if fused:
                f0 = self.fnets[0]
                wst = (self.fnets[1]["w1Fb"].data_ptr() - f0["w1Fb"].data_ptr()) // 2
                assert wst > 0 and all((self.fnets[k][key].data_ptr() - f0[key].data_ptr()) // 2 == k * wst
                                       for k in range(2 * ncoup) for key in ("w2Fb", "w1Fb", "w0Fb"))
                z0r = self._buf("z0_rec", (R, dim))
                sg = getattr(self, "_flow_sign", None)
                if sg is None or sg.shape[1] != B:      # (activations kept by the second-generation kernel: signs from the tensors themselves)
                    sg = ops.flow_sign_bits(kept[0], kept[1], B)
                ops.flow_reverse_chain(x_out, g_x, g_logp, -1.0 / N_all if g_logp is not None else 0.0, fl.mask, kept[2], sg,
                                       f0["w2Fb"], f0["w1Fb"], f0["w0Fb"], wst, GOb_all, G2b_all, G1b_all, XPb_all, Gc, f0["db2"],
                                       self.fnets[1]["rb2"] - f0["rb2"], z0r)
                # (the kernel leaves the per-image sums as [image][column] rows only; both bf16 operands of the conditioning layer's
                # reverse - the rows and their transpose - come from one pack launch instead of a scattered second layout + two casts)
                self._Gc_packed = ops.pack_transpose_bf16(Gc, out=self._buf("Gcond_b", (B, cstride), bf), outT=self._buf("GcondT_b", (cstride, B), bf))
                x_cur = z0r
            if not fused:
                self._need_fallback()                  # the coupling-by-coupling pass reads the plain operand layouts
            for i in range(ncoup - 1, -1, -1) if not fused else ():
                m = fl.mask[i]
                if grouped:
                    XPb, GOb, G2b_n, G1b_n = XPb_all[i], [GOb_all[2 * i], GOb_all[2 * i + 1]], [G2b_all[2 * i], G2b_all[2 * i + 1]], [G1b_all[2 * i], G1b_all[2 * i + 1]]
                ops.flow_mask_pad_mixed(x_cur, m, out_bf16=XPb)
                if kept is not None:                   # written out by the forward kernel (mhe_flow_couplings_bf16_emit)
                    H1b, H2b, O = [[k[2 * i + n] for n in range(2)] for k in kept]
                for n in range(2 if kept is None else 0):
                    d, slot = self.fnets[2 * i + n], (2 * i + n) * 2
                    ops.conv2d_nhwc(v4(XPb), d["w0b"], 1, 1, 1, 0, out=v4(P0b))
                    ops.flow_cond_lrelu_mixed(P0b, cflat[:, slot * h:], cstride, B, out_bf16=H1b[n])
                    ops.conv2d_nhwc(v4(H1b[n]), d["w1b"], 1, 1, 1, 0, out=v4(P2b))
                    ops.flow_cond_lrelu_mixed(P2b, cflat[:, (slot + 1) * h:], cstride, B, out_bf16=H2b[n])
                    ops.linear_bf16_f32out(H2b[n], d["w2b"], d["b2"], out=O[n])            # s, t pre-activations: f32 result, as the forward kernel
                x_in, g_in = (xa, ga) if x_cur is not xa else (xb, gb)
                ops.flow_couple_bwd(x_cur, O[0], O[1], m, g_cur, g_logp, -1.0 / N_all if g_logp is not None else 0.0, B, x_in, GO[0], GO[1], gpart,
                                    GOb[0], GOb[1], db_s=self.fnets[2 * i]["db2"], db_t=self.fnets[2 * i + 1]["db2"])
                for n in range(2):
                    d, slot = self.fnets[2 * i + n], (2 * i + n) * 2
                    if grouped:
                        G2b, G1b = G2b_n[n], G1b_n[n]
                    else:
                        ops.conv_wgrad(v4(H2b[n]), v4(GOb[n]), 1, 1, 1, 0, d["dw2"])
                    ops.conv2d_nhwc(v4(GOb[n]), d["w2Tb"], 1, 1, 1, 0, out=v4(P2b))
                    ops.flow_lrelu_bwd_sum(P2b, H2b[n], N, B, Gc[:, (slot + 1) * h:], Gc.shape[1], out_bf16=G2b, sum_out_t=GcT[(slot + 1) * h:])
                    if not grouped:
                        ops.conv_wgrad(v4(H1b[n]), v4(G2b), 1, 1, 1, 0, d["dw1"])
                    ops.conv2d_nhwc(v4(G2b), d["w1Tb"], 1, 1, 1, 0, out=v4(GH1b))
                    ops.flow_lrelu_bwd_sum(GH1b, H1b[n], N, B, Gc[:, slot * h:], Gc.shape[1], out_bf16=G1b, sum_out_t=GcT[slot * h:])
                    if not grouped:
                        ops.conv_wgrad(v4(XPb), v4(G1b), 1, 1, 1, 0, d["dw0"])
                    ops.linear_bf16_f32out(G1b, d["w0Tb"], out=GX[n])
                ops.flow_couple_accum(gpart, GX[0], GX[1], m, g_in)
                x_cur, g_cur = x_in, g_in
            if grouped:
                f0, nets = self.fnets[0], 2 * ncoup
                stride = self.fnets[1]["r1"] - f0["r1"]                   # the nets' raw-gradient slots are laid out at one pitch
                assert all(self.fnets[k][key] - f0[key] == k * stride for k in range(nets) for key in ("r0", "r1", "r2"))
                ops.conv_wgrad_batched(kept[1], GOb_all, f0["dw2"], stride, nets)           # dW2 = GO^T H2   [64, h]  x 24
                ops.conv_wgrad_batched(kept[0], G2b_all, f0["dw1"], stride, nets)           # dW1 = G2^T H1   [h, h]   x 24
                for n in range(2):      # dW0 = G1^T XP [h, 64]: the s (t) nets of the 12 couplings share their coupling's masked input
                    ops.conv_wgrad_batched(XPb_all, G1b_all[n], self.fnets[n]["dw0"], 2 * stride, ncoup, gy_batch_stride=2 * R * h)
            self.z0_recovered = x_cur
            return Gc
        for i in range(ncoup - 1, -1, -1):
            m = fl.mask[i]
            ops.flow_mask_pad(x_cur, m, XP)
            for n in range(2):
                d, slot = self.fnets[2 * i + n], (2 * i + n) * 2
                ops.linear(XP, d["w0"], out=Hb[n][0])
                if mixed:
                    ops.flow_cond_lrelu_mixed(Hb[n][0], cflat[:, slot * h:], cstride, B, out_bf16=H1b[n])
                    ops.conv2d_nhwc(v4(H1b[n]), d["w1b"], 1, 1, 1, 0, out=v4(P2b))
                    ops.flow_cond_lrelu_mixed(P2b, cflat[:, (slot + 1) * h:], cstride, B, out_f32=Hb[n][1])
                else:
                    ops.flow_cond_lrelu(Hb[n][0], cflat[:, slot * h:], cstride, B)
                    ops.linear(Hb[n][0], d["w1"], out=Hb[n][1])
                    ops.flow_cond_lrelu(Hb[n][1], cflat[:, (slot + 1) * h:], cstride, B)
                ops.linear(Hb[n][1], d["w2"], d["b2"], out=O[n])
            x_in, g_in = (xa, ga) if x_cur is not xa else (xb, gb)
            ops.flow_couple_bwd(x_cur, O[0], O[1], m, g_cur, g_logp, -1.0 / N_all if g_logp is not None else 0.0, B, x_in, GO[0], GO[1], gpart)
            for n in range(2):
                d, slot = self.fnets[2 * i + n], (2 * i + n) * 2
                ops.linear_wgrad(Hb[n][1], GO[n], d["dw2"]); ops.colsum(GO[n], d["db2"])
                ops.linear(GO[n], d["w2T"], out=G2)
                if mixed:
                    ops.flow_lrelu_bwd_mixed(G2, Hb[n][1], out_f32=G2, out_bf16=G2b)
                    ops.conv_wgrad(v4(H1b[n]), v4(G2b), 1, 1, 1, 0, d["dw1"])
                else:
                    ops.flow_lrelu_bwd(G2, Hb[n][1])
                    ops.linear_wgrad(Hb[n][0], G2, d["dw1"])
                ops.sum_over_hypotheses(G2, N, B, out=Gc[:, (slot + 1) * h:], out_stride=Gc.shape[1])
                if mixed:
                    ops.conv2d_nhwc(v4(G2b), d["w1Tb"], 1, 1, 1, 0, out=v4(GH1b))
                    ops.flow_lrelu_bwd_mixed(GH1b, H1b[n], out_f32=G1)
                else:
                    ops.linear(G2, d["w1T"], out=G1); ops.flow_lrelu_bwd(G1, Hb[n][0])
                ops.linear_wgrad(XP, G1, d["dw0"])
                ops.sum_over_hypotheses(G1, N, B, out=Gc[:, slot * h:], out_stride=Gc.shape[1])
                ops.linear(G1, d["w0T"], out=GX[n])
            ops.flow_couple_accum(gpart, GX[0], GX[1], m, g_in)
            x_cur, g_cur = x_in, g_in
        self.z0_recovered = x_cur
        return Gc

    # ------------------------------------------------------------------ the step
    def forward(self, x, y, noise=None, N=None, trunk_out=None):
        """forward of MHEnt.get_loss (hand/network.py:760-831) keeping what the reverse pass needs.  Returns the get_loss dict.
        trunk_out (B, feat_dim) f32 (testing aid): stands in for the ResNet trunk's output, whose forward and reverse
        passes are then skipped - the reference's golden gradients are pinned from the trunk feature on."""
        m = self.model
        N = N or m.loss_N
        B = x.shape[0] if trunk_out is None else trunk_out.shape[0]
        self.sync()           # someone else (torch.optim, load_state_dict) may have written the parameters
        f = self._trunk_forward(x.contiguous()) if trunk_out is None else trunk_out.contiguous()
        feat, feat_b = ops.linear(f, self.l1["w"], self.l1["b"], want_bf16=True) if self.flow_bf16 else (ops.linear(f, self.l1["w"], self.l1["b"]), None)
        hs, B_own, N_all, feat_own = None, B, N, feat
        if self.shard_hypotheses:
            if self.glow is not None:
                raise NotImplementedError("hypothesis sharding is wired for the RealNVP branch")
            from .dist import HypothesisShards
            hs = HypothesisShards(self.dist, N)
            feat_own, feat = feat, hs.gather_rows(feat)                       # (world*B, 512): every image's conditioning feature
            feat_b = None
            y = {"crop_uv": hs.gather_rows(y["crop_uv"]), "vis": hs.gather_rows(y["vis"])}
            if noise is not None:
                noise = hs.gather_hypothesis_rows(noise.reshape(N * B, 45), B)
            lo, hi = hs.hypotheses()
            N, B = hi - lo, feat.shape[0]                                      # local hypotheses x all images
        hd = ops.linear(feat, self.d0["w"], self.d0["b"], relu=True)
        det = ops.linear(hd, self.d2["w"], self.d2["b"])[:, :16].contiguous()
        fl = self.flow
        if self.glow is not None:
            if noise is not None and noise.dim() == 3:          # the reference's (B,N,45) layout -> sample-major rows
                noise = noise.permute(1, 0, 2).reshape(N * B, 45)
            z0 = m._noise(N * B, 1.0, noise, self.dev)
            cond = None
            th45, log_q = self.glow.forward(z0, feat)
        else:
            h, ncoup = fl.hidden, len(fl.mask)
            if self.f_wcb is not None:
                if feat_b is None:
                    feat_b = feat.to(torch.bfloat16)
                cond = ops.linear_bf16_f32out(feat_b, self.f_wcb, self.f_bc).view(B, 2 * ncoup, 2, h)
            else:
                cond = ops.linear(feat, self.f_wc, self.f_bc).view(B, 2 * ncoup, 2, h)
            z0 = m._noise(N * B, 1.0, noise, self.dev)
            self._flow_kept, self._flow_sign = None, None
            if self.flow_bf16 and h == 512 and os.environ.get("MHE_FLOW_RECOMPUTE") != "1":
                # the 512-wide kernel writes the nets' activations out on the way: the reverse pass reads them instead of re-evaluating
                # the nets coupling by coupling (what autograd would have kept)
                Rr = N * B
                kept = (self._buf("fl_h1", (2 * ncoup, Rr, h), torch.bfloat16), self._buf("fl_h2", (2 * ncoup, Rr, h), torch.bfloat16),
                        self._buf("fl_o", (2 * ncoup, Rr, 64)))
                f0 = self.fnets[0]
                if (os.environ.get("MHE_FLOW_FRAG", "1") == "1" and "f1F" in f0 and N % 64 == 0          # (the tape form needs whole 64-row chunks)
                        and ops.flow_couplings_frag_supported(Rr, B, z0.shape[1], h, ncoup)):
                    wst = (self.fnets[1]["f1F"].data_ptr() - f0["f1F"].data_ptr()) // 2
                    sg = self._buf("fl_sign", (2 * ncoup, Rr // 64, 2, 8, 64, 2), torch.int32)
                    th45, _, log_q = ops.flow_couplings_frag(z0, cond, f0["f0F"], f0["f1F"], f0["f2F"], wst, self.f_b2, fl.mask, B, h,
                                                             ops.FLOW_FORWARD, emit=kept, sign_bits=sg)
                    self._flow_sign = sg
                else:
                    self._need_fallback()              # the second-generation kernel's stream
                    th45, _, log_q = ops.flow_couplings_emit(z0, cond, self.f_stream, self.f_b2, fl.mask, B, h, ops.FLOW_FORWARD, *kept)
                self._flow_kept = kept
            else:
                self._need_fallback()
                th45, _, log_q = ops.flow_couplings(z0, cond, self.f_stream, self.f_b2, fl.mask, B, h, ops.FLOW_FORWARD)
        blob = m.mano_dec.table_blob()
        cu, vis = y["crop_uv"].contiguous(), y["vis"].contiguous()
        o = ops.mano_joints(th45, det, blob, cu, vis, m.b_2d, m.th45_ref_alpha, want=("log_p", "norms"))
        q_log_p, hq, log_p = ops.elbo_reduce(o["log_p"], log_q if m.entropy else None, N, B)
        if hs is not None:
            # means over the local hypotheses -> sums -> all-reduce -> means over all K; each rank reports its own images
            part = torch.stack([q_log_p, hq]) * (float(N) / N_all)
            hs.reduce_images(part)
            own = slice(hs.rank * B_own, (hs.rank + 1) * B_own)
            q_log_p, hq = part[0, own].contiguous(), part[1, own].contiguous()
            log_p = hq + q_log_p
        out = {"th_norm": o["norms"][:, 0], "bt_norm": o["norms"][:, 1], "q_log_p_z_giv_y": q_log_p,
               "log_p": log_p if m.entropy else q_log_p}
        if m.entropy:
            out["h_q_z_giv_i"] = hq
        self.tape = {"f": f, "feat": feat, "feat_b": feat_b, "hd": hd, "det": det, "cond": cond, "th45": th45, "blob": blob, "cu": cu, "vis": vis,
                     "N": N, "B": B, "trunk": trunk_out is None, "hs": hs, "B_own": B_own, "N_all": N_all, "feat_own": feat_own}
        return out

    def backward(self, g_log_p=None):
        """reverse pass of the last forward() for d loss / d log_p = g_log_p (B,) - default -1/B, the reference's
        total = mean_b(-log_p[b]) (hand/criteria.py:55,173); fills self.G."""
        m, t = self.model, self.tape
        N, B, f, feat, hd, det, cond, th45 = t["N"], t["B"], t["f"], t["feat"], t["hd"], t["det"], t["cond"], t["th45"]
        hs, B_own, N_all = t.get("hs"), t.get("B_own", B), t.get("N_all", N)
        self._wq = []                   # (a reverse pass that raised half way must not leave its queued weight gradients to the next one)
        self.raw.zero_()
        g_logp = self._buf("g_logp", (B,))
        if g_log_p is None:
            g_logp.fill_(-1.0 / B_own)          # per-rank mean over its own images; the ranks' gradients are averaged (/ world)
        elif hs is not None:
            g_logp.copy_(hs.gather_rows(g_log_p.reshape(B_own).contiguous()))
        else:
            g_logp.copy_(g_log_p.reshape(B))
        g45, gdet_rows = self._mano_bwd(th45, det, t["blob"], t["cu"], t["vis"], g_logp, N_all)
        if self.glow is not None:
            g_feat = self.glow.backward(g45, g_logp if m.entropy else None, N, B)
        else:
            Gc = self._flow_backward(th45, cond, g45, g_logp if m.entropy else None, N, B, N_all)
        # det head: gdet [B,16] -> padded [B,32]
        gdet = self._buf("gdet", (B, 32)); gdet.zero_()
        ops.sum_over_hypotheses(gdet_rows, N, B, out=gdet, out_stride=32)
        ops.linear_wgrad(hd, gdet, self.d2["dw"]); ops.colsum(gdet, self.d2["db"])
        ghd = ops.linear(gdet, self.d2["wT"]); ops.flow_lrelu_bwd(ghd, hd, slope=0.0)
        ops.linear_wgrad(feat, ghd, self.d0["dw"]); ops.colsum(ghd, self.d0["db"])
        if self.glow is None:       # conditioning projections of all nets in one pass
            # bf16 mode: both products of the conditioning projections take bf16 operands like the rest of the flow's reverse pass (f32
            # accumulation; the two f32 launches were 0.19 ms); the bias gradient sums the f32 Gc
            cond_bf16 = self.flow_bf16 and self.f_wcb is not None and B % 8 == 0 and self.cond_bwd_bf16
            # (a batch that is not a multiple of 8 cannot feed the bf16 kernel's 16-byte rows: f32 operands, the weights widened from
            # the bf16 copy for this call when the f32 copy is not kept)
            wc32 = self.f_wc if (cond_bf16 or self.cond_f32) else self.f_wcb.float()
            if cond_bf16:
                fb = t.get("feat_b")
                packed = getattr(self, "_Gc_packed", None)
                ops.linear_wgrad(fb if fb is not None else feat.to(torch.bfloat16), packed[0] if packed else Gc.to(torch.bfloat16), self.dwc)
            else:
                ops.linear_wgrad(feat, Gc, self.dwc)
            ops.colsum(Gc, self.dbc)
            if self.flow_bf16 and B % 4 == 0:
                # g_feat = Gc Wc is a (B x 24,576) x (24,576 x 512) product: 8 output tiles walking K serially as a plain GEMM
                # (~0.75 ms); as a split-K reduction over the 24,576 columns ("pixels" of the weight-gradient kernel, operands
                # GcT [k][b] and Wc [k][f] as they lie) it fills the chip
                K_, F_ = (self.f_wcb if self.f_wcb is not None else self.f_wc).shape
                g_feat = self._buf("g_feat_flow", (B, F_)); g_feat.zero_()
                if cond_bf16:
                    ops.conv_wgrad(self.f_wcb.view(K_, 1, 1, F_), (packed[1] if packed else self._GcT.to(torch.bfloat16)).view(K_, 1, 1, B),
                                   1, 1, 1, 0, g_feat)
                else:
                    GcT32 = self._GcT if getattr(self, "_Gc_packed", None) is None else Gc.t().contiguous()
                    ops.conv_wgrad(wc32.view(K_, 1, 1, F_), GcT32.view(K_, 1, 1, B), 1, 1, 1, 0, g_feat)
            else:
                g_feat = ops.linear(Gc, self.f_wcT if self.f_wcT is not None else wc32.t().contiguous())
        ops.add(g_feat, ops.linear(ghd, self.d0["wT"]))
        if hs is not None:          # partial over the local hypotheses, all images -> this rank's images, all hypotheses
            g_feat = hs.scatter_grad(g_feat)
        ops.linear_wgrad(f, g_feat, self.l1["dw"]); ops.colsum(g_feat, self.l1["db"])
        g_f = ops.linear(g_feat, self.l1["wT"])
        self._grad_ready(3)
        if t["trunk"]:
            self._trunk_backward(g_f)
        else:
            for i in (2, 1, 0):
                self._grad_ready(i)
        t.update({"g_feat": g_feat, "g_th45": g45, "g_trunk_out": g_f})
        if getattr(self.model, "_trainer", None) is self and self.world > 1:
            # autograd-bridge use under data parallelism: torch's optimizer expects the averaged gradient in .grad
            self.finish_allreduce()
            self.G.mul_(1.0 / self.world)
            self._G_averaged = True        # optimizer_step() must not divide by world a second time

    def forward_backward(self, x, y, noise=None, N=None, trunk_out=None):
        """forward + reverse pass of total = mean_b(-log_p[b]); fills self.G.  Returns the get_loss dict + 'total'."""
        out = self.forward(x, y, noise=noise, N=N, trunk_out=trunk_out)
        self.backward()
        out["total"] = -out["log_p"].mean()
        return out

    # ------------------------------------------------------------------ torch.autograd bridge
    def attach(self):
        """make `model.get_loss(...)` (training mode, grad enabled) differentiable: `total_loss.backward()` of the reference's
        loop (hand/CrossModalHand.py:455-470) then runs the hand-written reverse pass and leaves the gradients in every
        parameter's `.grad` (views of the flat gradient buffer), so the reference's own `clip_grad_norm_` and
        `torch.optim.Adam` work unchanged.  (The fused `step()` stays the fast path.)"""
        self.model._trainer = self
        return self

    def _mano_bwd(self, th45, det, blob, cu, vis, g_logp, N):
        R, B = th45.shape[0], det.shape[0]
        g45 = self._buf("g45", (R, 45)); rows = self._buf("gdet_rows", (R, 16))
        from . import _lib
        ops.check(_lib.lib().mhe_mano_joints_bwd_f32(ops._ptr(th45), ops._ptr(det), ops._ptr(cu), ops._ptr(vis), ops._ptr(blob), ops._ptr(g_logp),
                                                     ops._ptr(g45), ops._ptr(rows), R, B, float(self.model.b_2d), float(self.model.th45_ref_alpha),
                                                     1.0 / N, ops._stream()), "mhe_mano_joints_bwd_f32")
        return g45, rows

    def optimizer_step(self):
        """all-reduce (sum) over ranks, clip_grad_norm_(max_norm) and Adam in one fused pass"""
        self.finish_allreduce()          # the buckets were handed to RCCL as the reverse pass completed them
        ops.train_tick(self.step_t, self.sq)
        if self.max_norm and self.max_norm > 0:
            ops.sqnorm(self.G, self.sq)
        self.last_grad_scale = 1.0 if self._G_averaged else 1.0 / self.world
        ops.adam_step(self.P, self.G, self.M, self.V, self.sq, self.step_t, self.lr, self.betas[0], self.betas[1], self.eps,
                      self.max_norm or 0.0, self.last_grad_scale)
        self._G_averaged = False
        self.repack()          # every derived operand layout follows the new parameters
        if self.glow is not None:
            self.glow.invalidate()
        self._param_ver = sum(p._version for p in self._params)

    def second_bn_update(self):
        """what a SECOND train-mode encoder pass over the same batch does to the BatchNorm buffers (the reference's metrics
        pass, hand/CrossModalHand.py:355-361: identical batch statistics, so running = (1-m) running + m stat once more and
        num_batches_tracked + 1), from the statistics the forward kept - a handful of multi-tensor launches, no second pass"""
        units = [u for u in self.units if getattr(u, "mi", None) is not None]
        means = [u.mi[0] for u in units]
        var = torch._foreach_pow([u.mi[1] for u in units], -2.0)            # 1/invstd^2 = biased var + eps
        torch._foreach_sub_(var, BN_EPS)
        # pixels per channel of the unit's output (a folded conv3 - 1x1, stride 1 - never wrote it: its input has the same pixels)
        npix = [float(u.y.numel() // u.cout) if u.y is not None else float(u.x.numel() // u.cin) for u in units]
        torch._foreach_mul_(var, [n / max(n - 1.0, 1.0) for n in npix])
        torch._foreach_lerp_([u.bn.running_mean for u in units], means, BN_MOMENTUM)
        torch._foreach_lerp_([u.bn.running_var for u in units], var, BN_MOMENTUM)
        torch._foreach_add_([u.bn.num_batches_tracked for u in units], 1)

    def step(self, x, y, noise=None, N=None, test_samples=0, temp=0.8, double_bn_update=True):
        """one iteration of the reference's training loop (hand/CrossModalHand.py:353-361,455-470).  test_samples > 0
        adds its per-iteration metrics pass `sample(N=[n,n], temp=0.8, mods={uv,xyz,verts})` to the returned dict,
        from the conditioning feature of THIS forward.  The reference runs the encoder a second time on the same
        batch in train mode for it: same feature, but the BatchNorm running statistics advance twice per iteration -
        double_bn_update=True (default) reproduces that on the buffers, so checkpoints / eval-mode results match a
        reference-trained model."""
        out = self.forward_backward(x, y, noise=noise, N=N)
        if test_samples:
            if double_bn_update and self.tape["trunk"]:
                self.second_bn_update()
            with torch.no_grad():
                out.update(self.model.sample(None, N=[test_samples, test_samples], temp=temp, mods={"uv", "xyz", "verts"}, y=y,
                                             feat=self.tape["feat_own"]))
        self.optimizer_step()
        return out


class GraphedStep:
    """TrainStep.step replayed from HIP graphs (launch-bound: ~840 kernels per step).  One process: one graph.  Data parallel:
    the step is cut where a gradient bucket is complete - a graph ends there, the bucket's all-reduce is issued eagerly
    (RCCL, async_op: it runs on the communicator's stream under the next graph's kernels), the next graph starts; the last
    graph (norm, clip, Adam, operand re-pack) is launched after the waits.  Six graphs and four collectives per step."""

    def __init__(self, ts, x, y, noise=None, N=None, test_samples=0, criterion=None):
        """test_samples / criterion: the reference's whole iteration (hand/CrossModalHand.py:349-361,452-470) in the graph - the
        metrics pass sample(N=[n,n], temp=0.8) from this forward's feature and `criterion(out, y)` (MHEntLoss: 14 metrics);
        self.out then carries 'criterion' = (total, losses, metrics)"""
        if ts.shard_hypotheses:
            raise NotImplementedError("GraphedStep: the hypothesis-sharded forward has collectives inside the forward pass")
        self.ts, self.graphs, self.actions = ts, [], []
        _step = ts.step

        def step(x, y, noise=None, N=None):
            out = _step(x, y, noise=noise, N=N, test_samples=test_samples)
            if criterion is not None:
                with torch.no_grad():
                    out["criterion"] = criterion(dict(out), y)
            return out
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # allocations and lazy initialisation happen here, not under capture.  This warm-up IS one real optimizer step on the
            # batch held by x / y: `warm_out` is that iteration's result, and a training loop must not replay() the same batch
            # again (mhentropy_amd/run.py takes warm_out for the capture iteration; the reference steps once per iteration,
            # hand/CrossModalHand.py:455-470)
            self.warm_out = step(x, y, noise=noise, N=N)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self._mode = "thread_local" if ts.comm else "global"           # the communicator's watchdog thread may touch the device
        with torch.cuda.stream(side):
            self._cur = torch.cuda.CUDAGraph()
            self._cur.capture_begin(capture_error_mode=self._mode)
            ts._capture = self
            try:
                self.out = step(x, y, noise=noise, N=N)
            finally:
                ts._capture = None
                self._cur.capture_end()
            self.graphs.append(self._cur)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self._fb_in_graph = ts._fb_keep        # does the captured end-of-step repack refresh the flow's fallback layouts? (replay())

    def cut(self, action):
        self._cur.capture_end()
        self.graphs.append(self._cur)
        self.actions.append(action)
        self._cur = torch.cuda.CUDAGraph()
        self._cur.capture_begin(pool=self.graphs[0].pool(), capture_error_mode=self._mode)

    def replay(self):
        ts = self.ts
        if not self._fb_in_graph:              # the captured repack did not refresh the fallback layouts: they no longer follow the parameters
            ts._fb_stale = True
            ts._poison_fallback()
        for k, g in enumerate(self.graphs):
            g.replay()
            if k < len(self.actions):
                a = self.actions[k]
                if a[0] == "allreduce":
                    ts._all_reduce_bucket(a[1])
                elif ts._xchg is not None:
                    ts._xchg.finish()
        return self.out


class _LossFn(torch.autograd.Function):
    """MHEnt.get_loss as one autograd node: forward = TrainStep.forward, backward = TrainStep.backward(d loss / d log_p)."""
    @staticmethod
    def forward(ctx, trainer, x, y, N, noise, *params):
        out = trainer.forward(x, y, noise=noise, N=N)
        ctx.trainer, ctx.keys = trainer, list(out)
        vals = tuple(out[k] for k in ctx.keys)
        ctx.mark_non_differentiable(*[v for k, v in zip(ctx.keys, vals) if k != "log_p"])
        return vals

    @staticmethod
    def backward(ctx, *grads):
        tr = ctx.trainer
        g = grads[ctx.keys.index("log_p")]
        tr.backward(None if g is None else g.contiguous().float())
        return (None, None, None, None, None) + tuple(tr.grad_of(p) for p in tr._params)


def differentiable_get_loss(trainer, x, y, N=None, noise=None):
    vals = _LossFn.apply(trainer, x, y, N, noise, *trainer._params)
    return dict(zip(_LossFn_keys(trainer), vals))


def _LossFn_keys(trainer):
    return ["th_norm", "bt_norm", "q_log_p_z_giv_y", "log_p"] + (["h_q_z_giv_i"] if trainer.model.entropy else [])
