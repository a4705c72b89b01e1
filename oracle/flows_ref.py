"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the reference's conditional RealNVP
(/root/reference/hand/flows.py) as pure functions over a state_dict whose keys
are the reference's own (`mask`, `{s,t}.{i}.l.{0,1,2}.*`, `{s,t}.{i}.c.{0,1}.*`).
Parity is pinned by tests/golden/flow_*.npz, produced by importing the reference
in the build container (oracle/gen_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.
"""
import math
import torch
import torch.nn.functional as F


def coupling_net(sd, net, i, x, cond, num_hidden=2):
    """One s- or t-network.  reference flows.py:97-122 (`_nets.forward`):
    l0(x); then for each hidden layer: + c_j(cond), leaky_relu(0.01), l_{j+1};
    tanh on the s network only (flows.py:120-121)."""
    p = f"{net}.{i}."
    h = F.linear(x, sd[p + "l.0.weight"], sd[p + "l.0.bias"])
    for j in range(num_hidden):
        if cond is not None:
            h = h + F.linear(cond, sd[p + f"c.{j}.weight"], sd[p + f"c.{j}.bias"])
        h = F.leaky_relu(h)
        h = F.linear(h, sd[p + f"l.{j + 1}.weight"], sd[p + f"l.{j + 1}.bias"])
    if net == "s":
        h = torch.tanh(h)
    return h


def forward_p(sd, z, cond):
    """z -> x, the sampling direction.  reference flows.py:210-217."""
    mask = sd["mask"]
    x = z
    for i in range(mask.shape[0]):
        m = mask[i]
        x_ = x * m
        s = coupling_net(sd, "s", i, x_, cond) * (1 - m)
        t = coupling_net(sd, "t", i, x_, cond) * (1 - m)
        x = x_ + (1 - m) * (x * torch.exp(s) + t)
    return x


def forward_p_logdet(sd, z, cond):
    """forward_p that also returns sum_i sum_d s_i,d (the log-det of z -> x);
    SURVEY.md appendix A2(ii): log q(x) = logN(z0) - that sum."""
    mask = sd["mask"]
    x = z
    tot = z.new_zeros(z.shape[0])
    for i in range(mask.shape[0]):
        m = mask[i]
        x_ = x * m
        s = coupling_net(sd, "s", i, x_, cond) * (1 - m)
        t = coupling_net(sd, "t", i, x_, cond) * (1 - m)
        x = x_ + (1 - m) * (x * torch.exp(s) + t)
        tot = tot + s.sum(1)
    return x, tot


def backward_p(sd, x, cond):
    """x -> z with log|det|.  reference flows.py:219-227."""
    mask = sd["mask"]
    z = x
    log_det = x.new_zeros(x.shape[0])
    for i in reversed(range(mask.shape[0])):
        m = mask[i]
        z_ = m * z
        s = coupling_net(sd, "s", i, z_, cond) * (1 - m)
        t = coupling_net(sd, "t", i, z_, cond) * (1 - m)
        z = (1 - m) * (z - t) * torch.exp(-s) + z_
        log_det = log_det - s.sum(1)
    return z, log_det


def std_normal_logprob(z):
    """MultivariateNormal(0, I_d).log_prob, reference flows.py:157,320."""
    d = z.shape[1]
    return -0.5 * (z * z).sum(1) - 0.5 * d * math.log(2 * math.pi)


def log_prob(sd, x, feat):
    """reference flows.py:271-331 for tsfm_on=int, scale=1, weights=1 and dim>3
    (`make_cond` returns feat unchanged, flows.py:243-244,258-268)."""
    z, log_det = backward_p(sd, x, feat)
    return std_normal_logprob(z) + log_det


def sample(sd, z0, feat):
    """reference flows.py:333-359 with the prior draw `z0` supplied by the caller
    (already multiplied by temp); scale=1."""
    return forward_p(sd, z0, feat)


def _q(t):
    return t.bfloat16().float()


def coupling_net_bf16(sd, net, i, x, cond):
    """coupling_net with the rounding points of the product's bf16 performance mode
    (csrc/flow_bf16.hip): operands of every product rounded to bf16 (inputs, weights, hidden
    activations after leaky_relu), accumulation, biases, conditioning and activations in f32."""
    p = f"{net}.{i}."
    h = F.linear(_q(x), _q(sd[p + "l.0.weight"])) + (F.linear(cond, sd[p + "c.0.weight"], sd[p + "c.0.bias"]) + sd[p + "l.0.bias"])
    h = _q(F.leaky_relu(h))
    h = F.linear(h, _q(sd[p + "l.1.weight"])) + (F.linear(cond, sd[p + "c.1.weight"], sd[p + "c.1.bias"]) + sd[p + "l.1.bias"])
    h = _q(F.leaky_relu(h))
    o = F.linear(h, _q(sd[p + "l.2.weight"])) + sd[p + "l.2.bias"]
    return torch.tanh(o) if net == "s" else o


def forward_p_logdet_bf16(sd, z, cond):
    mask = sd["mask"]
    x = z
    tot = z.new_zeros(z.shape[0])
    for i in range(mask.shape[0]):
        m = mask[i]
        x_ = x * m
        s = coupling_net_bf16(sd, "s", i, x_, cond) * (1 - m)
        t = coupling_net_bf16(sd, "t", i, x_, cond) * (1 - m)
        x = x_ + (1 - m) * (x * torch.exp(s) + t)
        tot = tot + s.sum(1)
    return x, tot
