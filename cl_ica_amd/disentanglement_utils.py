"""Disentanglement scores on device tensors (/root/reference/disentanglement_utils.py:17-221).

Same two entry points, arguments and return structure as the reference:

    (score, corr_or_None), (z_2, hz_2)  = linear_disentanglement(z, hz, mode, train_test_split)
    (score, corr_or_None), Thz          = permutation_disentanglement(z, hz, mode, rescaling, solver, sign_flips, cache_permutations)

with every ``mode`` ("r2", "adjusted_r2", "pearson", "spearman") and both solvers ("naive", "munkres").  The reference copies the
4096 x n embeddings to the host and runs sklearn / numpy / scipy over them every ``n_log_steps``; here ONE pass of the HIP library
(``clica_moments``: G = [z | hz | 1]^T [z | hz | 1] in fp64, csrc/moments.hip) is the only thing that touches the data, and every score is
evaluated from that (2n + 1)^2 matrix in fp64 on the host (one launch covers 2n + 1 <= 129, i.e. n <= 64; wider inputs go through the
same kernel over 64-column blocks, ``ops.moments``.  Inputs are taken in fp32, the reference's embedding dtype; fp64 host arrays are
rounded to fp32 first -- the sums themselves are fp64):

  * LinearRegression (:97-100): normal equations on G's hz / 1 block; r2_score of the prediction (:23) from the quadratic form
    z^T z - 2 c^T X^T z + c^T X^T X c -- no second pass over the data; Pearson of (z, prediction) likewise (a linear map of G);
  * np.corrcoef (:40) / the per-latent rescaling beta_j = <z_j, hz_j> / <hz_j, hz_j> (:158): entries of G;
  * spearmanr (:38) = Pearson of the average ranks: ranks by a device sort (torch), then the same moment pass on the ranks;
  * the naive solver's n! 2^n candidate matrices (:165-216) are scored from G, not by n! 2^n passes over the data;
  * the munkres solver's assignment on -|corr| (:43-45): scipy's Hungarian on the n x n host matrix (same optimum as munkres.py).

Reference quirks kept on purpose (a drop-in must return what the reference returns): with ``rescaling=True`` the candidate
transformation T is overwritten by the rescaled identity (:150-160: ``Thz = X @ beta``), so the naive solver scores the SAME
matrix for every permutation; ``max`` then returns the first candidate.  mode r2 / adjusted_r2 require the naive solver (:130).
"""
from __future__ import annotations

import itertools

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import lazy, ops

__all__ = ["linear_disentanglement", "permutation_disentanglement"]

_MODES = ("r2", "adjusted_r2", "pearson", "spearman")


def _dev32(x, device=None):
    x = lazy.plain(x) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
    x = x.detach()
    if device is None:      # host data goes to the GPU (there is no CPU path: ops.moments refuses CPU tensors)
        device = x.device if (x.is_cuda or not torch.cuda.is_available()) else torch.device("cuda", torch.cuda.current_device())
    return x.to(device=device, dtype=torch.float32)


def _average_ranks(x: torch.Tensor) -> torch.Tensor:
    """scipy.stats.rankdata(method="average") per column, on the device (what spearmanr correlates)."""
    M, n = x.shape
    order = torch.argsort(x, dim=0, stable=True)
    sx = torch.gather(x, 0, order)
    pos = torch.arange(1, M + 1, device=x.device, dtype=torch.float64).unsqueeze(1).expand(M, n)
    new = torch.ones_like(sx, dtype=torch.bool)
    new[1:] = sx[1:] != sx[:-1]
    grp = torch.cumsum(new.to(torch.int64), 0) - 1                       # tie-group id per sorted position and column
    flat = grp + torch.arange(n, device=x.device).unsqueeze(0) * M        # unique across columns
    tot = torch.zeros(M * n, dtype=torch.float64, device=x.device).index_add_(0, flat.reshape(-1), pos.reshape(-1))
    cnt = torch.zeros(M * n, dtype=torch.float64, device=x.device).index_add_(0, flat.reshape(-1), torch.ones(M * n, dtype=torch.float64, device=x.device))
    avg = (tot / cnt.clamp_min(1))[flat]
    ranks = torch.empty((M, n), dtype=torch.float64, device=x.device)
    ranks.scatter_(0, order, avg)
    return ranks.to(torch.float32)                                        # ranks <= 2^24 are exact in fp32 (M <= 16.7 M)


class _Moments:
    """Blocks of G = [z | h | 1]^T [z | h | 1] (host fp64) for M samples."""

    def __init__(self, z: torch.Tensor, h: torch.Tensor):
        self.M, self.a, self.b = z.shape[0], z.shape[1], h.shape[1]
        G = ops.moments(z, h).cpu().numpy()
        a, b = self.a, self.b
        self.zz, self.zh, self.hh = G[:a, :a], G[:a, a:a + b], G[a:a + b, a:a + b]
        self.sz, self.sh = G[:a, -1], G[a:a + b, -1]

    def transformed(self, T: np.ndarray, c: np.ndarray = None):
        """Moments of (z, h @ T + c): returns (z^T p, p^T p diag-capable full matrix, sum p)."""
        c = np.zeros(T.shape[1]) if c is None else c
        zp = self.zh @ T + np.outer(self.sz, c)
        pp = T.T @ self.hh @ T + np.outer(T.T @ self.sh, c) + np.outer(c, T.T @ self.sh) + self.M * np.outer(c, c)
        sp = T.T @ self.sh + self.M * c
        return zp, pp, sp


def _score(mo: _Moments, zp, pp, sp, mode: str):
    """_disentanglement(z, p, mode, reorder=False) (:17-58) from moments: p has as many columns as z."""
    M, a = mo.M, mo.a
    if mode in ("r2", "adjusted_r2"):
        ss_res = np.diag(mo.zz) - 2.0 * np.diag(zp) + np.diag(pp)
        ss_tot = np.diag(mo.zz) - mo.sz ** 2 / M
        r2 = float(np.mean(1.0 - ss_res / ss_tot))                        # sklearn r2_score, uniform average
        if mode == "adjusted_r2":
            r2 = 1.0 - (1.0 - r2) * (M - 1) / (M - a - 1)
        return r2, None
    cov = zp - np.outer(mo.sz, sp) / M
    vz = np.diag(mo.zz) - mo.sz ** 2 / M
    vp = np.diag(pp) - sp ** 2 / M
    corr = cov / np.sqrt(np.outer(vz, vp))
    return float(np.mean(np.abs(np.diag(corr)))), corr


def linear_disentanglement(z, hz, mode="r2", train_test_split=False):
    """Disentanglement up to linear transformations (:63-102): fit LinearRegression hz -> z on the first half (or everything),
    score the prediction on the second half (or everything)."""
    if mode not in _MODES:
        raise AssertionError(f"mode {mode!r}")
    z = _dev32(z)
    hz = _dev32(hz, z.device)
    if train_test_split:
        k = len(z) // 2
        z1, h1, z2, h2 = z[:k], hz[:k], z[k:], hz[k:]
    else:
        z1, h1, z2, h2 = z, hz, z, hz
    fit = _Moments(z1, h1)
    b = fit.b
    XtX = np.block([[fit.hh, fit.sh[:, None]], [fit.sh[None, :], np.array([[float(fit.M)]])]])
    XtZ = np.vstack([fit.zh.T, fit.sz[None, :]])
    coef = np.linalg.lstsq(XtX, XtZ, rcond=None)[0]                        # (b + 1, a): weights and intercept
    W, c = coef[:b], coef[b]
    pred = ops.linear_fwd(h2.contiguous(), torch.as_tensor(W.T.copy(), dtype=torch.float32, device=z.device),
                          torch.as_tensor(c, dtype=torch.float32, device=z.device), leaky=False, slope=0.0)
    if mode == "spearman":                                                 # rank correlation of (z_2, prediction): needs the ranks of both
        mo = _Moments(_average_ranks(z2), _average_ranks(pred))
        return _score(mo, mo.zh, mo.hh, mo.sh, "pearson"), (z2, pred)
    ev = fit if not train_test_split else _Moments(z2, h2)
    zp, pp, sp = ev.transformed(W, c)
    return _score(ev, zp, pp, sp, mode), (z2, pred)


def _signed_permutations(n: int, sign_flips: bool):
    """The candidate matrices of the naive solver in the reference's generation order (:165-203): row by row, columns ascending,
    sign +1 before -1."""
    signs = (1.0, -1.0) if sign_flips else (1.0,)

    def rec(row, used, T):
        if row == n:
            yield T.copy()
            return
        for col in range(n):
            if col in used:
                continue
            for sg in signs:
                T[row, col] = sg
                yield from rec(row + 1, used | {col}, T)
                T[row, col] = 0.0
    yield from rec(0, frozenset(), np.zeros((n, n)))


def permutation_disentanglement(z, hz, mode="r2", rescaling=True, solver="naive", sign_flips=True, cache_permutations=None):
    """Disentanglement up to permutations (:105-221), by the Munkres assignment or by trying every (signed) permutation."""
    assert solver in ("naive", "munkres")
    if mode in ("r2", "adjusted_r2"):
        assert solver == "naive", "R2 coefficient is only supported with naive solver"
    if mode not in _MODES:
        raise AssertionError(f"mode {mode!r}")
    z = _dev32(z)
    hz = _dev32(hz, z.device)
    n = z.shape[-1]
    if rescaling:
        assert z.shape == hz.shape
    raw = _Moments(z, hz)
    beta = (np.diag(raw.zh) / np.diag(raw.hh)) if rescaling else None       # :158
    # Spearman = Pearson of the average ranks.  A positive column scale leaves ranks alone and a negative one maps rank r to
    # M + 1 - r, which negates the correlation exactly -- so the ranks of hz are taken ONCE and every candidate (rescaled identity,
    # signed permutation) is a column permutation of them plus sign flips of columns of the correlation matrix.
    rank_mo = _Moments(_average_ranks(z), _average_ranks(hz)) if mode == "spearman" else None

    def rank_score(T):
        P = (T != 0).astype(np.float64)
        zp, pp, sp = rank_mo.transformed(P)
        _, corr = _score(rank_mo, zp, pp, sp, "pearson")
        corr = corr * np.where(T.sum(0) < 0, -1.0, 1.0)[None, :]
        return float(np.mean(np.abs(np.diag(corr)))), corr

    def evaluate(T):
        """test_transformation(T, reorder=False) (:147-161): returns ((score, corr), matrix applied to hz)."""
        Teff = np.diag(beta) if rescaling else T                             # the reference overwrites T @ hz by the rescaled hz
        if mode == "spearman":
            return rank_score(Teff), Teff
        zp, pp, sp = raw.transformed(Teff)
        return _score(raw, zp, pp, sp, mode), Teff

    if solver == "munkres":
        (score, corr), Teff = evaluate(np.eye(n))
        rows, cols = linear_sum_assignment(-np.abs(corr))                   # Munkres on -|corr| (:43-45)
        perm = cols[np.argsort(rows)]
        Tp = Teff[:, perm]
        if mode == "spearman":
            best = rank_score(Tp)
        else:
            zp, pp, sp = raw.transformed(Tp)
            best = _score(raw, zp, pp, sp, mode)
        Tbest = Teff                       # (the reference returns the rescaled latents BEFORE the reordering, :160-161 / :221)
    else:
        if cache_permutations:
            cache = permutation_disentanglement.__dict__.setdefault("permutation_matrices", {})
            key = (rescaling, n, bool(sign_flips))
            if key not in cache:
                cache[key] = list(_signed_permutations(n, sign_flips))
            cands = cache[key]
        else:
            cands = _signed_permutations(n, sign_flips)
        best, Tbest = None, None
        for T in cands:
            res, Teff = evaluate(T)
            if best is None or res[0] > best[0]:                            # max(): the first of equal scores wins
                best, Tbest = res, Teff
            if rescaling:
                break                                                       # every candidate scores the same matrix (see the module docstring)
    thz = ops.linear_fwd(hz.contiguous(), torch.as_tensor(Tbest.T.copy(), dtype=torch.float32, device=hz.device), None, leaky=False, slope=0.0)
    return best, thz
