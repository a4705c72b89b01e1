"""ResNet-18/50 trunk driven through the HIP implicit-GEMM convolution.

Stands where the reference uses torchvision (`models.resnet18/50`, then
`res.fc = nn.Identity()`, reference hand/network.py:54-61).  Parameters and buffers
carry torchvision's names (conv1, bn1, layer{1-4}.{i}.conv{1-3}/bn{1-3}/downsample.{0,1})
so a state_dict saved by the reference loads unchanged; the nn.Conv2d /
nn.BatchNorm2d objects are parameter holders only - their forward is never
called.  Activations are NHWC; BatchNorm + ReLU of a producer is applied by the
consumer while it loads its operand, the tail of each residual block is one
fused elementwise pass.
"""
import torch
from torch import nn

from . import ops

CFG = {
    "resnet18": ("basic", (2, 2, 2, 2), 512),
    "resnet50": ("bottleneck", (3, 4, 6, 3), 2048),
}
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def pack_conv_weight(w, dtype, cin_pad=None):
    """torch [Cout,Cin,KH,KW] -> [Cout, Kpad] with k = (kh,kw,cin), cin fastest,
    channels zero-padded to cin_pad and K to the kernel's 128-byte stage."""
    Cout, Cin, KH, KW = w.shape
    cin_pad = cin_pad or Cin
    bke = 32 if dtype == torch.float32 else 64
    wp = torch.zeros(Cout, KH, KW, cin_pad, dtype=torch.float32, device=w.device)
    wp[..., :Cin] = w.detach().float().permute(0, 2, 3, 1)
    k = KH * KW * cin_pad
    kpad = (k + bke - 1) // bke * bke
    out = torch.zeros(Cout, kpad, dtype=torch.float32, device=w.device)
    out[:, :k] = wp.reshape(Cout, k)
    return out.to(dtype).contiguous()


def pack_stem_weight(w, dtype):
    """torch [64,3,7,7] -> [64,192] for the fused stem kernel: k = 24*kh + 3*kw + c (zero-padded)."""
    wp = torch.zeros(64, 8, 24, dtype=torch.float32, device=w.device)
    wp[:, :7, :21] = w.detach().float().permute(0, 2, 3, 1).reshape(64, 7, 21)
    return wp.reshape(64, 192).to(dtype).contiguous()


class _Block(nn.Module):
    def __init__(self, kind, inplanes, planes, stride):
        super().__init__()
        self.kind, self.stride = kind, stride
        exp = 4 if kind == "bottleneck" else 1
        if kind == "bottleneck":
            self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False); self.bn3 = nn.BatchNorm2d(planes * 4)
        else:
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
        if stride != 1 or inplanes != planes * exp:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * exp, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * exp))
        else:
            self.downsample = None


class ResNetTrunk(nn.Module):
    def __init__(self, arch="resnet50", compute_dtype=torch.float32):
        super().__init__()
        kind, blocks, self.feat_dim = CFG[arch]
        self.arch, self.kind = arch, kind
        self.compute_dtype = compute_dtype
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes, exp = 64, (4 if kind == "bottleneck" else 1)
        for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), blocks)):
            layer = []
            for bi in range(nb):
                layer.append(_Block(kind, inplanes, planes, 2 if (bi == 0 and li > 0) else 1))
                inplanes = planes * exp
            setattr(self, f"layer{li + 1}", nn.Sequential(*layer))
        self.fc = nn.Identity()       # the reference overwrites fc with Identity (network.py:61)
        self._wcache = {}
        # Producer BN+ReLU: "load" applies it in the consumer's operand load (no extra HBM pass, but the
        # consumer repeats it once per output-channel tile and per 3x3 tap), "pass" applies it once in
        # place with an elementwise kernel.  Measured on MI355X (tools/conv_bench.py): the in-place pass is
        # cheaper in both dtypes (bf16 15.2 -> 13.6 ms, f32 38.9 -> 35.7 ms per C2 step); "load" stays selectable.
        import os
        self.bn_apply = os.environ.get("MHE_BN_APPLY", "pass")
        # the 1x1 conv3 of a bottleneck reads its operand exactly once per output-channel tile: there the on-load form saves the
        # in-place pass over y2 without the 9-tap repetition a 3x3 consumer would pay, but its register-staged main loop is 12-14 us
        # per launch behind the LDS-DMA kernel the in-place form can use.  Measured per layer at C2 (tools/conv_variants.py
        # --bn-load): the saved pass is worth 56 / 28 / 14 / 7 us on layer1..4 -> "auto" = on load where the bottleneck is <= 128
        # wide (layer1, layer2), in place on layer3 / layer4 (C2 bf16 forward 9.19 ms all in place, 9.02 all on load)
        self.bn_apply_1x1 = os.environ.get("MHE_BN_APPLY_1X1", "auto")
        self.bn_apply_3x3 = os.environ.get("MHE_BN_APPLY_3X3", "auto")
        # (stride-2 3x3 consumers: each input element is used by 2.25 taps on average instead of 9: on load, -0.06 ms at C2)
        # up to this many channels (0 = never): at layer4.0's 512 the in-place pass + phase-pipelined kernel stay ahead (113 against 166 us)
        self.bn_load_s2 = int(os.environ.get("MHE_BN_LOAD_S2", "256"))
        # stride-1 3x3 consumers on 32 x 32 / 16 x 16 images (conv2 of layer2 / layer3 at C2): the kernel that keeps the input tile in LDS
        # and normalises every element once on its way in (csrc/conv_halo.hip) - no pass, no im2col re-reads
        self.conv_halo = os.environ.get("MHE_CONV_HALO", "1") == "1"
        # evaluate relu(bn3(conv3) + identity) inside the next block's conv1 (one read of the block output saved)
        self.fuse_tail = os.environ.get("MHE_FUSE_TAIL", "1") == "1"
        # ... and at 64 / 128 bottleneck channels (layer1 / layer2) do not write conv3's raw output at all: a statistics-only launch gives
        # bn3's batch statistics, the tail kernel evaluates conv3 again on its way (csrc/conv_fuse.hip; bf16 storage, forward-only path -
        # the train step keeps the raw outputs for its reverse pass)
        self.fuse_recompute = os.environ.get("MHE_FUSE_RECOMPUTE", "1") == "1"
        # bn3's batch statistics for those blocks: "gram" = from the Gram matrix of conv3's input (csrc/conv_gram.hip: the product is not
        # evaluated at all), "stream" = the statistics-only launch of the streaming kernel (evaluates, rounds and sums the product)
        self.recompute_stats = os.environ.get("MHE_RECOMPUTE_STATS", "gram")
        # the stem's max pool taken inside the conv1 kernel on the raw output (bf16, 256x256 images; csrc/stem_pool.hip): the
        # full-resolution conv1 output is never written, layer1.0's conv1 / shortcut apply bn1 + ReLU on their operand load
        self.stem_pool_fused = os.environ.get("MHE_STEM_POOL", "1") == "1"
        # the last block's relu(bn(y) + identity) evaluated inside the global average pool (csrc/conv.hip: bn_act_avgpool_kernel)
        self.fuse_pool = os.environ.get("MHE_FUSE_POOL", "1") == "1"

    # -- packed-weight cache keyed on the parameter's version counter
    def _w(self, conv, cin_pad=None, stem=False):
        p = conv.weight
        ext = getattr(self, "_external_w", None)
        if ext is not None and id(p) in ext:        # kept current on the device by train.TrainStep
            return ext[id(p)]
        key = (id(p), p._version, self.compute_dtype, p.device)
        hit = self._wcache.get(id(p))
        if hit is None or hit[0] != key:
            hit = (key, pack_stem_weight(p, self.compute_dtype) if stem else pack_conv_weight(p, self.compute_dtype, cin_pad))
            self._wcache[id(p)] = hit
        return hit[1]

    def _w_halo(self, conv):
        """the 3x3 weights as csrc/conv_halo.hip streams them (cached like _w; None while a trainer owns the packs: they change every step)"""
        if getattr(self, "_external_w", None) is not None:
            return None
        p = conv.weight
        key = (id(p), p._version, "halo", p.device)
        hit = self._wcache.get(("halo", id(p)))
        if hit is None or hit[0] != key:
            hit = (key, ops.conv3x3_halo_pack(self._w(conv)))
            self._wcache[("halo", id(p))] = hit
        return hit[1]

    def _bn_affine(self, y, bn, st, count=None):
        """this layer's BatchNorm folded to (scale, shift): batch statistics in training, running ones in eval"""
        if self.training:
            count = count if count is not None else y.numel() // y.shape[-1]
            # the finalize launch also clears the accumulators it read (self-cleaning arena) and counts the batch
            return ops.bn_finalize(st, bn.weight, bn.bias, bn.running_mean, bn.running_var, count, BN_MOMENTUM, BN_EPS, clear=True,
                                   num_batches_tracked=bn.num_batches_tracked)
        sc = bn.weight.detach() / torch.sqrt(bn.running_var + BN_EPS)
        return sc.contiguous(), (bn.bias.detach() - bn.running_mean * sc).contiguous()

    def _conv_bn(self, x, conv, bn, stats_pool, in_aff=None, stride=1, pad=0, k=1, cin_pad=None, apply=None):
        """raw conv output + this layer's BatchNorm folded to (scale, shift)."""
        w = self._w(conv, cin_pad)
        if in_aff is not None and (apply or self.bn_apply) == "pass":
            x = ops.bn_act(x, in_aff[0], in_aff[1], relu=True, out=x)
            in_aff = None
        isc, ish = in_aff if in_aff is not None else (None, None)
        st = stats_pool.take(conv.out_channels) if self.training else None
        y = ops.conv2d_nhwc(x, w, k, k, stride, pad, in_scale=isc, in_shift=ish, relu_in=in_aff is not None, stats=st)
        return y, self._bn_affine(y, bn, st)

    def forward(self, x):
        """x (B,3,H,W) float32 NCHW -> (B, feat_dim) float32."""
        dt = self.compute_dtype
        if getattr(self, "_external_sync", None) is not None:
            self._external_sync()             # trainer-owned operand packs follow the parameters (optimizer.step, load_state_dict)
        pool = self._stats_pool(x.device) if self.training else None
        st = pool.take(64) if self.training else None
        blocks = [blk for li in range(4) for blk in getattr(self, f"layer{li + 1}")]
        a_aff = None            # BatchNorm + ReLU still to be applied to `a` by its consumers (the fused stem leaves the pooled RAW output)
        if (self.stem_pool_fused and blocks[0].kind == "bottleneck" and blocks[0].downsample is not None
                and ops.stem_pool_supported(x.shape[0], x.shape[2], x.shape[3], dt)):
            a = ops.stem_conv7x7s2_pool(x.contiguous(), self._w(self.conv1, stem=True), self.bn1.weight.detach(), stats=st)
            a_aff = self._bn_affine(None, self.bn1, st, count=x.shape[0] * 128 * 128)
        else:
            y = ops.stem_conv7x7s2(x.contiguous(), self._w(self.conv1, stem=True), dt, stats=st)      # reads the NCHW image directly
            aff = self._bn_affine(y, self.bn1, st)
            a = ops.maxpool3x3s2(y, aff[0], aff[1])
        pending = None          # (raw conv3 output, bn3 affine, identity tensor, identity affine | None): an unevaluated block tail
        for bi, blk in enumerate(blocks):
            if blk.kind == "bottleneck":
                if pending is not None and isinstance(pending[0], str):
                    # ... with the previous block's conv3 evaluated again inside the same kernel (its raw output was never written)
                    _, y2_p, a2_p, w3_p, al_p, idt_p, idaff_p = pending
                    st = pool.take(blk.conv1.out_channels) if self.training else None
                    a, y1 = ops.bottleneck_tail(y2_p, a2_p, w3_p, al_p, idt_p, idaff_p, self._w(blk.conv1), stats=st)
                    a1 = self._bn_affine(y1, blk.bn1, st)
                    pending = None
                elif pending is not None:
                    # the previous block's relu(bn3(y3) + identity) is evaluated inside this conv1's operand load,
                    # which also writes it out once as this block's identity
                    yl_p, al_p, idt_p, idaff_p = pending
                    a = torch.empty_like(yl_p)
                    st = pool.take(blk.conv1.out_channels) if self.training else None
                    y1 = ops.conv1x1_residual_in(yl_p, idt_p, self._w(blk.conv1), al_p[0], al_p[1],
                                                 None if idaff_p is None else idaff_p[0],
                                                 None if idaff_p is None else idaff_p[1], a_out=a, stats=st)
                    a1 = self._bn_affine(y1, blk.bn1, st)
                    pending = None
                else:
                    y1, a1 = self._conv_bn(a, blk.conv1, blk.bn1, pool, a_aff, apply="load")
                # 3x3 consumer: in place, except where the row-streaming kernel runs (layer1 at C2) - it normalises each input row once on
                # its way into LDS (95 us against 56 + 77 us for the pass and the plain form)
                ap2 = "load" if self.bn_apply_3x3 == "auto" and (ops.conv_tile_choice(
                    y1.shape[0], y1.shape[1], y1.shape[2], y1.shape[3], blk.conv2.out_channels, 3, blk.stride, 1, y1.dtype, 1) == 9
                    or (blk.stride == 2 and blk.conv2.in_channels <= self.bn_load_s2)) else None
                wh = None
                if (self.conv_halo and blk.stride == 1 and y1.dtype == torch.bfloat16 and self.bn_apply_3x3 == "auto"
                        and ops.conv3x3_halo_supported(y1.shape[0], y1.shape[1], y1.shape[2], y1.shape[3], blk.conv2.out_channels)):
                    wh = self._w_halo(blk.conv2)
                if wh is not None:
                    st2 = pool.take(blk.conv2.out_channels) if self.training else None
                    y2 = ops.conv3x3_halo(y1, wh, a1[0], a1[1], relu_in=True, stats=st2)
                    a2 = self._bn_affine(y2, blk.bn2, st2)
                else:
                    y2, a2 = self._conv_bn(y1, blk.conv2, blk.bn2, pool, a1, blk.stride, 1, 3, apply=ap2)
                # (... and where the resident-slab kernel runs conv3, layer3 at C2: its transfer waves normalise each K tile once)
                ap3 = self.bn_apply_1x1 if self.bn_apply_1x1 != "auto" else ("load" if blk.conv3.in_channels <= 128 or ops.conv_tile_choice(
                    y2.shape[0], y2.shape[1], y2.shape[2], y2.shape[3], blk.conv3.out_channels, 1, 1, 0, y2.dtype, 1) == 11 else "pass")
                nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
                recompute = (self.fuse_recompute and self.fuse_tail and nxt is not None and nxt.kind == "bottleneck" and y2.dtype == torch.bfloat16
                             and nxt.conv1.kernel_size == (1, 1) and nxt.conv1.stride == (1, 1)
                             and ops.bottleneck_tail_supported(y2.shape[0], y2.shape[1], y2.shape[2], y2.shape[3], nxt.conv1.out_channels))
                if recompute:
                    w3 = self._w(blk.conv3)
                    st3 = None
                    if self.training and self.recompute_stats == "gram":
                        bn3 = blk.bn3
                        yl, al = None, ops.conv1x1_gram_bn(y2, a2[0], a2[1], w3, bn3.weight, bn3.bias, bn3.running_mean, bn3.running_var,
                                                           pool.gram(y2.shape[-1]), BN_MOMENTUM, BN_EPS, num_batches_tracked=bn3.num_batches_tracked)
                    else:
                        if self.training:    # bn3's batch statistics from the products as they would be stored - nothing is stored
                            st3 = pool.take(blk.conv3.out_channels)
                            ops.conv1x1_stats(y2, w3, a2[0], a2[1], st3)
                        yl, al = None, self._bn_affine(None, blk.bn3, st3, count=y2.numel() // y2.shape[-1])
                else:
                    yl, al = self._conv_bn(y2, blk.conv3, blk.bn3, pool, a2, apply=ap3)
            else:
                recompute = False
                y1, a1 = self._conv_bn(a, blk.conv1, blk.bn1, pool, None, blk.stride, 1, 3)
                yl, al = self._conv_bn(y1, blk.conv2, blk.bn2, pool, a1, 1, 1, 3)
            if blk.downsample is not None:
                idt, idaff = self._conv_bn(a, blk.downsample[0], blk.downsample[1], pool, a_aff, blk.stride, 0, 1, apply="load")
                a_aff = None        # (only the first block sees the un-normalised pooled stem output)
            else:
                idt, idaff = a, None
            nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
            if recompute:
                pending = ("re", y2, a2, w3, al, idt, idaff)
            elif self.fuse_tail and nxt is not None and nxt.kind == "bottleneck":
                pending = (yl, al, idt, idaff)
            elif nxt is None and self.fuse_pool and yl.shape[-1] % 4 == 0:
                # the last block's tail inside the average pool (same bits; no block-wide store: MHE_FUSE_POOL=0 keeps the two launches)
                if pool is not None:
                    pool.done()
                return self.fc(ops.bn_act_avgpool(yl, al[0], al[1], idt, None if idaff is None else idaff[0], None if idaff is None else idaff[1]))
            elif idaff is not None:
                a = ops.bn_act(yl, al[0], al[1], idt, idaff[0], idaff[1], relu=True)
            else:
                a = ops.bn_act(yl, al[0], al[1], idt, relu=True)
        if pool is not None:
            pool.done()
        return self.fc(ops.avgpool(a))

    def _stats_pool(self, device):
        """the trunk's statistics arena: allocated (zeroed) once; every slice handed out is cleared again by the bn_finalize launch
        that consumes it, so a forward starts on zeros without a fill launch"""
        p = getattr(self, "_pool", None)
        if p is None or p.buf.device != device:
            p = self._pool = _StatsPool(device, persistent=True)
        p.begin()
        return p


class _StatsPool:
    """one zeroed arena for all sharded per-channel (sum, sum^2) accumulators of a pass.  persistent=True: kept across passes by its
    owner - every slice must then be consumed by a launch that clears it (ops.bn_finalize(clear=True)); begin() / done() bracket a
    pass, and a pass that did not finish (an exception in between) leaves the arena marked dirty: the next begin() zeroes it."""
    def __init__(self, device, channels=32768, persistent=False):
        self.S = ops.stat_shards()
        self.buf = torch.zeros(2 * self.S * 2 * channels, device=device, dtype=ops.STAT_DTYPE)      # fixed-point words (include/mhe.h: mhe_stat_t)
        self.off = 0
        self.persistent, self.clean = persistent, True
        self._gram = {}

    def gram(self, Cb):
        """the Gram accumulators + f64 workspace for bottleneck width Cb (ops.conv1x1_gram_bn; self-cleaning like the arena)"""
        if Cb not in self._gram:
            self._gram[Cb] = ops.gram_buffers(Cb, self.buf.device)
        return self._gram[Cb]

    def begin(self):
        if not self.clean:
            self.buf.zero_()
            for g, _ in self._gram.values():
                g.zero_()
        self.off, self.clean = 0, False

    def done(self):
        self.clean = True

    def take(self, C):
        n = 2 * self.S * 2 * C
        if self.off + n > self.buf.numel():
            if self.persistent:
                raise RuntimeError("_StatsPool: persistent arena exhausted")
            self.buf = torch.zeros_like(self.buf)
            self.off = 0
        v = self.buf[self.off:self.off + n].view(2, self.S, 2, C)
        self.off += n
        self.high = max(getattr(self, "high", 0), self.off)          # (high-water mark: what a pass has touched)
        return v


def resnet18(pretrained=False, **kw):
    return ResNetTrunk("resnet18", **kw)


def resnet50(pretrained=False, **kw):
    return ResNetTrunk("resnet50", **kw)
