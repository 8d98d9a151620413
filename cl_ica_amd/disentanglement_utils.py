"""Disentanglement scores used by the training driver's periodic evaluation, computed on the GPU
(SURVEY.md section 8(f) row N2 -- the step either side of the hot path).

Same entry points and return structure as /root/reference/disentanglement_utils.py for the two calls
the drivers make (main_mlp.py:218-231, 336-352):
  linear_disentanglement(z, hz, mode="r2")                      -> ((r2, None), (z, hz_pred))
  permutation_disentanglement(z, hz, mode="pearson", solver="munkres", rescaling=True)
                                                                -> ((mcc, corr), Thz)
The reference moves 4096 x n samples to the host for sklearn + a pure-Python Hungarian solver; here the
regression and correlation run as torch device ops and only the n x n correlation matrix goes to the
host for the assignment (scipy.optimize.linear_sum_assignment).
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

__all__ = ["linear_disentanglement", "permutation_disentanglement"]


def _t(x):
    return x.detach().to(torch.float64) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x), dtype=torch.float64)


def _r2(z, pred):
    """sklearn.metrics.r2_score default (uniform average over outputs), disentanglement_utils.py:23-24."""
    ss_res = ((z - pred) ** 2).sum(0)
    ss_tot = ((z - z.mean(0, keepdim=True)) ** 2).sum(0)
    return float((1.0 - ss_res / ss_tot).mean())


def linear_disentanglement(z, hz, mode="r2", train_test_split=False):
    """R^2 of the best affine map hz -> z (LinearRegression with intercept, disentanglement_utils.py:63-102)."""
    if mode != "r2":
        raise NotImplementedError("only mode='r2' is used by the drivers")
    z, hz = _t(z), _t(hz).to(_t(z).device)
    if train_test_split:
        k = len(z) // 2
        z1, h1, z2, h2 = z[:k], hz[:k], z[k:], hz[k:]
    else:
        z1, h1, z2, h2 = z, hz, z, hz
    ones = torch.ones(len(h1), 1, dtype=h1.dtype, device=h1.device)
    X = torch.cat([h1, ones], 1)
    # normal equations: n <= 64, 4096 samples -- well conditioned in fp64
    coef = torch.linalg.solve(X.T @ X, X.T @ z1)
    pred = torch.cat([h2, torch.ones(len(h2), 1, dtype=h2.dtype, device=h2.device)], 1) @ coef
    return (_r2(z2, pred), None), (z2, pred)


def permutation_disentanglement(z, hz, mode="r2", rescaling=True, solver="naive", sign_flips=True, cache_permutations=None):
    """Mean correlation coefficient up to permutation (disentanglement_utils.py:105-221, munkres branch)."""
    if mode != "pearson" or solver != "munkres":
        raise NotImplementedError("only mode='pearson', solver='munkres' is used by the drivers")
    z, hz = _t(z), _t(hz).to(_t(z).device)
    dim = z.shape[-1]
    if rescaling:      # per-latent least-squares scale (does not change |Pearson|, kept for the returned Thz)
        beta = (z * hz).sum(0) / (hz ** 2).sum(0)
        thz = hz * beta
    else:
        thz = hz
    zc = z - z.mean(0, keepdim=True)
    hc = thz - thz.mean(0, keepdim=True)
    corr = (zc.T @ hc) / torch.sqrt((zc ** 2).sum(0)[:, None] * (hc ** 2).sum(0)[None, :])
    c = corr.cpu().numpy()
    rows, cols = linear_sum_assignment(-np.abs(c))     # == Munkres on -|corr| (:43-45)
    perm = cols[np.argsort(rows)]
    c_sorted = c[:, perm]
    return (float(np.abs(np.diag(c_sorted)).mean()), c_sorted), thz[:, torch.as_tensor(perm, device=thz.device)]
