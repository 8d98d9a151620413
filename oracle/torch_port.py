"""CPU baseline: the reference's unsupervised train step restated in plain PyTorch CPU ops.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/np_oracle.py header).  This is the
``cpu_baseline`` leg of bench.py (kind "port"): the same op sequence the reference executes --
broadcast subtract -> torch.norm(p) -> pow(p) -> cat -> logsumexp (losses.py:447-467), two
nn.Sequential passes of B rows (main_mlp.py:270-271), roll inside the graph (:272),
torch.optim.Adam (:312) -- timed on the host cores of the GPU box.  /root/reference itself does not
travel to the box.  Pinned like the NumPy oracle: tests/test_oracle_golden.py checks this port
against the goldens as well.
"""
from __future__ import annotations

import time
from typing import List

import torch
from torch import nn


def lp_simclr_loss(z1, z2, z3, p, tau=1.0, alpha=0.5, compat=True, pow=True):
    """losses.py:443-477 (p >= 1 branch) in torch ops, materialising (B,B3,n) like the reference."""
    neg = torch.norm(z1.unsqueeze(1) - z3.unsqueeze(0), p=p, dim=-1)
    pos = torch.norm(z1 - z2, p=p, dim=-1)
    if pow:
        neg, pos = neg.pow(p), pos.pow(p)
    if compat:
        lse = torch.logsumexp(-torch.cat((neg, pos.unsqueeze(1)), dim=1) / tau, dim=1)
    else:
        lse = torch.logsumexp(-neg / tau, dim=1) - torch.log(torch.tensor(float(neg.shape[1])))
    loss_pos = pos / tau
    loss = 2 * (alpha * loss_pos + (1.0 - alpha) * lse)
    return loss.mean(), loss, [loss_pos.mean(), lse.mean()]


def make_mlp(n: int, hidden: List[int]) -> nn.Sequential:
    """encoders.py:36-48: Linear + LeakyReLU(0.01) per hidden layer, final Linear."""
    dims = [n] + list(hidden) + [n]
    mods: List[nn.Module] = []
    for i in range(len(dims) - 1):
        mods.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            mods.append(nn.LeakyReLU())
    return nn.Sequential(*mods)


def make_mixing(n: int, n_layers: int = 3) -> nn.Sequential:
    mods: List[nn.Module] = []
    for i in range(n_layers):
        lin = nn.Linear(n, n, bias=False)
        w = torch.rand(n, n) * 2 - 1
        lin.weight.data = w / w.norm(dim=0, keepdim=True)
        mods.append(lin)
        if i < n_layers - 1:
            mods.append(nn.LeakyReLU(0.2))
    g = nn.Sequential(*mods)
    for prm in g.parameters():
        prm.requires_grad = False
    return g


def sample_box(B, n, sigma=0.05):
    """spaces.py:273-302 + spaces_utils.py:106-142 (element-wise truncated resampling)."""
    z = torch.rand(B, n)
    out = torch.full((B, n), float("nan"))
    done = ~torch.isnan(out)
    while int(done.sum()) < B * n:
        cand = torch.randn(B, n) * sigma + z
        ok = (cand >= 0) & (cand <= 1) & ~done
        out[ok] = cand[ok]
        done |= ok
    return z, out


def time_reference_step(n=10, B=6144, p=2, steps=5, warmup=2, lr=1e-4, threads=None):
    """Median seconds per unsupervised step on the host CPU (sample -> g -> f x2 -> loss -> backward -> Adam)."""
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(0)
    f = make_mlp(n, [n * 10, n * 50, n * 50, n * 50, n * 50, n * 10])
    g = make_mixing(n)
    opt = torch.optim.Adam(f.parameters(), lr=lr)
    times, last = [], None
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        z1, z2 = sample_box(B, n)
        opt.zero_grad()
        a = f(g(z1)); b = f(g(z2)); c = torch.roll(a, 1, 0)
        tot, _, _ = lp_simclr_loss(a, b, c, p)
        tot.backward()
        opt.step()
        last = tot.item()
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], last
