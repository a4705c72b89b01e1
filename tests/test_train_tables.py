"""CPU: the host-side index tables of the train step (mhentropy_amd/train.py) - pure data-movement logic that every
derived weight layout and the data-gradient convolutions depend on; checked against torch's own conv / autograd on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mhentropy_amd import ops, train


@pytest.mark.parametrize("k,stride,pad", [(1, 1, 0), (3, 1, 1), (3, 2, 1), (1, 2, 0)])
def test_dgrad_operand_index_gives_the_input_gradient(k, stride, pad):
    """a convolution of the (zero-dilated) output gradient with W'[ci][kh'][kw'][co] = W[co][ci][KH-1-kh'][KW-1-kw'] at padding k-1-pad
    is the input gradient - the identity train.conv_dgrad builds on the forward kernel"""
    g = torch.Generator().manual_seed(k * 10 + stride)
    Cin, Cout, H = 6, 5, 8
    x = torch.randn(2, Cin, H, H, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    idx = train.dgrad_operand_index(torch.arange(w.numel()).view(w.shape))          # [Cin, (kh', kw', co)]
    wd = w.reshape(-1)[idx].view(Cin, k, k, Cout).permute(0, 3, 1, 2)               # as a torch conv weight [Cin, Cout, k, k]
    if stride == 2:                                                                 # zero-dilate: out[2i, 2j] = gy[i, j]
        d = torch.zeros(2, Cout, H, H, dtype=torch.float64)
        d[:, :, ::2, ::2] = gy
    else:
        d = gy
    if k == 1 and stride == 2:          # computed on the coarse grid and scattered (train.conv_dgrad's third form)
        gx = torch.zeros_like(x)
        gx[:, :, ::2, ::2] = F.conv2d(gy, wd)
    else:
        gx = F.conv2d(d, wd, padding=k - 1 - pad)
    assert (gx - x.grad).abs().max() < 1e-10


def test_parity_split_stride2_dgrad_tables():
    """the four parity-class operands of a 3x3 / stride-2 / pad-1 data gradient (train.dgrad_s2_operand_indices, consumed by
    mhe_conv3x3s2_dgrad_nhwc): output pixel (2i+py, 2j+px) = a (1+py) x (1+px)-tap convolution of gy, taps at offsets 0 / +1,
    reads past the edge are zero - restated with torch convs and checked against autograd"""
    g = torch.Generator().manual_seed(5)
    Cin, Cout, H = 6, 5, 10
    x = torch.randn(2, Cin, H, H, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64)
    y = F.conv2d(x, w, stride=2, padding=1)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    tabs = train.dgrad_s2_operand_indices(torch.arange(w.numel()).view(w.shape))
    gx = torch.zeros_like(x)
    for py in range(2):
        for px in range(2):
            tb = tabs[2 * py + px]
            wd = w.reshape(-1)[tb].view(Cin, 1 + py, 1 + px, Cout).permute(0, 3, 1, 2)      # torch conv weight [Cin, Cout, th, tw]
            gp = F.pad(gy, (0, px, 0, py))                                              # zeros past the bottom / right edge
            gx[:, :, py::2, px::2] = F.conv2d(gp, wd)
    assert (gx - x.grad).abs().max() < 1e-10
    used = torch.cat([t.reshape(-1) for t in tabs])
    assert used.numel() == w.numel() and torch.equal(torch.sort(used)[0], torch.arange(w.numel()))      # every weight exactly once


@pytest.mark.parametrize("bf16", [False, True])
def test_flow_stream_table_reproduces_the_host_packer(bf16):
    """gathering [W0|W1|W2] through the table == running the host packer on the weights (the table is obtained by packing
    index-valued weights; for the bf16 stream the index travels as three base-128 digits)"""
    dim, h = 45, 128
    rng = np.random.default_rng(0)
    w0, w1, w2 = (rng.normal(size=s).astype(np.float32) for s in ((h, dim), (h, h), (dim, h)))
    if bf16:        # values exactly representable in bf16, so the packer's rounding is the identity
        w0, w1, w2 = ((torch.from_numpy(a).bfloat16().float().numpy()) for a in (w0, w1, w2))
    tab = train.flow_stream_table(dim, h, bf16)
    flat = np.concatenate([w0.ravel(), w1.ravel(), w2.ravel()])
    got = np.where(tab >= 0, flat[np.maximum(tab, 0)], 0.0).astype(np.float32)
    if bf16:
        want = (ops.flow_pack_net_bf16(w0, w1, w2).astype(np.uint32) << 16).view(np.float32)
    else:
        want = ops.flow_pack_net(w0, w1, w2)
    assert got.shape == want.shape and np.array_equal(got, want)
    used = tab[tab >= 0]
    assert used.size == flat.size and np.array_equal(np.sort(used), np.arange(flat.size))       # every weight exactly once
