"""CPU oracle: a NumPy restatement of the cl-ica contrastive hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cl_ica_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker.  The product path is the HIP library behind ``include/clica.h``.

Parity pin: the reference ships no tests, so this oracle is pinned
against golden vectors produced by importing the reference itself in the build container
(``tests/golden/gen_goldens.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``).

Each function cites the reference lines it restates.  Arithmetic is float64 by default so the
oracle is a tighter target than the fp32 reference (reference fp32 vs this fp64 restatement:
loss rel <= 1e-7, grads rel <= 3e-7 on the goldens).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

__all__ = [
    "lp_pair_matrix", "lp_simclr_loss", "simclr_loss", "mlp_forward", "mlp_backward",
    "rescale_head", "softclip_head", "adam_step", "mixing_forward", "train_step",
    "MLPParams",
]

_CHUNK = 256  # rows per block of the pair matrix (bounds memory at B=6144: 256*6144*n*8 B)


def _chunk_rows(cols: int, n: int) -> int:
    """Rows per block so that one (rows, cols, n) fp64 temporary stays below ~512 MB (pools of 49 152 x 40)."""
    return int(max(1, min(_CHUNK, (1 << 26) // max(1, cols * n))))


# --------------------------------------------------------------------------------------
# LpSimCLRLoss  (reference: losses.py:405-477, helper _logmeanexp losses.py:506-510)
# --------------------------------------------------------------------------------------
def _lse(x: np.ndarray, axis: int) -> np.ndarray:
    m = np.max(x, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    return (np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True)) + m).squeeze(axis)


def _powabs(d: np.ndarray, p: float) -> np.ndarray:
    a = np.abs(d)
    if p == 1:
        return a
    if p == 2:
        return d * d
    if p == 3:
        return a * a * a
    return a ** p


def _dpowabs(d: np.ndarray, p: float) -> np.ndarray:
    """d/dd |d|^p, with the sub-gradient at 0 taken as 0 (torch.norm backward masks it;
    SURVEY.md section 7 'Exact zeros')."""
    if p == 1:
        return np.sign(d)
    if p == 2:
        return 2.0 * d
    a = np.abs(d)
    with np.errstate(divide="ignore", invalid="ignore"):
        g = p * np.where(a > 0, a ** (p - 1.0), 0.0) * np.sign(d)
    return g


def lp_pair_matrix(z1: np.ndarray, z3: np.ndarray, p: float, pow: bool = True) -> np.ndarray:
    """neg[i, j] = ||z1_i - z3_j||_p (optionally ** p).  losses.py:447-454 (p >= 1 branch)."""
    d = z1[:, None, :] - z3[None, :, :]
    s = _powabs(d, p).sum(-1)
    return s if pow else s ** (1.0 / p)


def lp_simclr_loss(
    z1: np.ndarray, z2: np.ndarray, z3: np.ndarray, p: float, tau: float = 1.0, alpha: float = 0.5,
    compat: bool = False, pow: bool = True, grad: bool = True,
    g_mean: float = 1.0, g_item: Optional[np.ndarray] = None, g_pos: float = 0.0, g_neg: float = 0.0,
    dtype=np.float64,
) -> Dict[str, np.ndarray]:
    """LpSimCLRLoss.loss forward (+ analytic backward).

    Forward restates losses.py:430-477; the backward is the autograd of that graph
    (norm_backward masked at 0 o pow_backward).  ``g_*`` are upstream gradients for the four
    autograd-connected outputs ``(mean, per_item, [pos_mean, neg_mean])``.

    The p < 1 branch (losses.py:433-442) builds the negatives in the TRANSPOSED orientation
    neg[j, i] = || |z1_i - z3_j + 1e-12| ||_p, so row r of the loss is anchored at z3_r.
    """
    z1 = np.asarray(z1, dtype); z2 = np.asarray(z2, dtype); z3 = np.asarray(z3, dtype)
    B, n = z1.shape
    frac = p < 1.0
    if frac:
        # rows of the pair matrix are z3 rows; torch.cat((neg, pos[:, None]), 1) needs B3 == B
        if compat:
            assert z3.shape[0] == B, "p<1 with compat mode needs B3 == B (torch.cat, losses.py:459)"
        rows, cols, sgn, eps = z3, z1, -1.0, 1e-12
    else:
        rows, cols, sgn, eps = z1, z3, 1.0, 0.0
    R, C = rows.shape[0], cols.shape[0]
    inv_p = 1.0 / p

    # positive pair, losses.py:439-441 / :450
    dpos = z1 - z2
    if frac:
        apos = np.abs(dpos) + 1e-12        # eps OUTSIDE abs for the positive (losses.py:441)
        spos = (apos ** p).sum(-1)
    else:
        spos = _powabs(dpos, p).sum(-1)
    pos = spos if pow else spos ** inv_p
    assert pos.shape[0] == R or not compat

    lse = np.empty(R, dtype)
    negs = []
    _CHUNK = _chunk_rows(C, rows.shape[1])
    for r0 in range(0, R, _CHUNK):
        d = sgn * (rows[r0:r0 + _CHUNK, None, :] - cols[None, :, :]) + eps   # eps INSIDE abs (losses.py:436)
        s = _powabs(d, p).sum(-1)
        neg = s if pow else s ** inv_p
        x = -neg / tau
        if compat:
            x = np.concatenate([x, (-pos[r0:r0 + _CHUNK] / tau)[:, None]], axis=1)   # losses.py:459-462
            lse[r0:r0 + _CHUNK] = _lse(x, 1)
        else:
            lse[r0:r0 + _CHUNK] = _lse(x, 1) - math.log(C)                            # losses.py:463-465,506-510
        negs.append(None)
    loss_pos = pos / tau
    loss_i = 2.0 * (alpha * loss_pos + (1.0 - alpha) * lse)                           # losses.py:467
    out = dict(loss_mean=loss_i.mean(), loss_i=loss_i, pos_mean=loss_pos.mean(), neg_mean=lse.mean(),
               lse=lse, pos=pos)
    if not grad:
        return out

    Rn = loss_i.shape[0]
    gi = g_mean / Rn + (0.0 if g_item is None else np.asarray(g_item, dtype))
    A = 2.0 * alpha * gi + g_pos / Rn               # coefficient on pos_i / tau
    Cc = 2.0 * (1.0 - alpha) * gi + g_neg / Rn      # coefficient on lse_i
    A = np.broadcast_to(A, (Rn,)).astype(dtype); Cc = np.broadcast_to(Cc, (Rn,)).astype(dtype)
    lse_raw = lse if compat else lse + math.log(C)

    d_rows = np.zeros_like(rows); d_cols = np.zeros_like(cols)
    for r0 in range(0, R, _CHUNK):        # _CHUNK: the adaptive block size chosen above
        sl = slice(r0, r0 + _CHUNK)
        d = sgn * (rows[sl, None, :] - cols[None, :, :]) + eps
        s = _powabs(d, p).sum(-1)
        neg = s if pow else s ** inv_p
        w = np.exp(-neg / tau - lse_raw[sl, None])                 # softmax weight of entry j in row i
        coef = -(Cc[sl, None] / tau) * w                           # dL/dneg_ij
        if not pow:
            with np.errstate(divide="ignore", invalid="ignore"):
                coef = coef * np.where(s > 0, inv_p * s ** (inv_p - 1.0), 0.0)
        g = coef[:, :, None] * _dpowabs(d, p)                      # dL/dd_ijk
        d_rows[sl] += sgn * g.sum(1)
        d_cols -= sgn * g.sum(0)
    # positive-pair term
    if compat:
        wpos = np.exp(-pos / tau - lse_raw)
        cpos = A / tau - Cc * wpos / tau
    else:
        cpos = A / tau
    if not pow:
        with np.errstate(divide="ignore", invalid="ignore"):
            cpos = cpos * np.where(spos > 0, inv_p * spos ** (inv_p - 1.0), 0.0)
    if frac:
        gp = cpos[:, None] * p * apos ** (p - 1.0) * np.sign(dpos)
    else:
        gp = cpos[:, None] * _dpowabs(dpos, p)
    if frac:
        dz3, dz1 = d_rows, d_cols
    else:
        dz1, dz3 = d_rows, d_cols
    out.update(dz1=dz1 + gp, dz2=-gp, dz3=dz3)
    return out


def lp_symmetric_row_grads(z1_rows, z2_rows, pool, lse_rows, lse_pool, p, tau=1.0, alpha=0.5, local_rows=None,
                           dtype=np.float64):
    """Gradient of the data-parallel objective w.r.t. a set of LOCAL rows when the negatives pool is "all z1_rec of the global
    batch" (main_mlp.py:272 z3_rec = roll(z1_rec)): with d_ij = d_ji the row part (weights w_ij = softmax of
    row i) and the column part (w_ji = softmax of row j, from every rank) collapse into one sum over the pool,
        dz1_i = gp_i - (C / tau) sum_j (w_ij + w_ji) d neg_ij / d z1_i ,    C = 2 (1 - alpha) / B_local   (compat mode, pow=True)
    given the log-sum-exp of the rows (``lse_rows``) and of every pool row (``lse_pool``).  Returns (dz1, dz2) for the rows.
    Used to check the engine's symmetric backward sweep at pool sizes where the full fp64 pair matrix is out of reach."""
    z1 = np.asarray(z1_rows, dtype); z2 = np.asarray(z2_rows, dtype); P = np.asarray(pool, dtype)
    Ls = np.asarray(lse_rows, dtype); Lp = np.asarray(lse_pool, dtype)
    Bl = z1.shape[0] if local_rows is None else int(local_rows)
    A = 2.0 * alpha / Bl; Cc = 2.0 * (1.0 - alpha) / Bl
    dz1 = np.zeros_like(z1)
    ch = _chunk_rows(P.shape[0], z1.shape[1])
    for r0 in range(0, z1.shape[0], ch):
        sl = slice(r0, r0 + ch)
        d = z1[sl, None, :] - P[None, :, :]
        neg = _powabs(d, p).sum(-1)
        w = np.exp(-neg / tau - Ls[sl, None]) + np.exp(-neg / tau - Lp[None, :])
        dz1[sl] = ((-(Cc / tau) * w)[:, :, None] * _dpowabs(d, p)).sum(1)
    dpos = z1 - z2
    pos = _powabs(dpos, p).sum(-1)
    cpos = A / tau - Cc * np.exp(-pos / tau - Ls) / tau
    gp = cpos[:, None] * _dpowabs(dpos, p)
    return dz1 + gp, -gp


# --------------------------------------------------------------------------------------
# SimCLRLoss  (reference: losses.py:162-202)
# --------------------------------------------------------------------------------------
def uniformity_loss(z1, z3, p=2.0, grad=True, dtype=np.float64):
    """UniformityLoss.loss (losses.py:211-222): pair tensor z1[None] - z3[:, None] -> lp[j, i]; per item j
    logsumexp_i(-lp[j, i]) - log(#z1 rows) (_logmeanexp, losses.py:506-510); mean over j.  Backward = autograd."""
    z1 = np.asarray(z1, dtype); z3 = np.asarray(z3, dtype)
    d = z1[None, :, :] - z3[:, None, :]                 # [j, i, k]
    lp = _powabs(d, p).sum(-1)
    lse = _lse(-lp, 1)
    item = lse - np.log(z1.shape[0])
    out = dict(loss_mean=item.mean(), loss_i=item)
    if grad:
        w = np.exp(-lp - lse[:, None]) / z3.shape[0]    # d mean / d(-lp[j, i])
        gd = -w[:, :, None] * _dpowabs(d, p)            # d mean / d d[j, i, k]
        out["dz1"] = gd.sum(0)
        out["dz3"] = -gd.sum(1)
    return out


def alignment_loss(z1, z2, p=2.0, grad=True, dtype=np.float64):
    """AlignmentLoss.loss (losses.py:231-241): per item sum_k |z1 - z2|^p, mean over items."""
    z1 = np.asarray(z1, dtype); z2 = np.asarray(z2, dtype)
    d = z1 - z2
    item = _powabs(d, p).sum(-1)
    out = dict(loss_mean=item.mean(), loss_i=item)
    if grad:
        g = _dpowabs(d, p) / z1.shape[0]
        out["dz1"], out["dz2"] = g, -g
    return out


def simclr_loss(z1, z2, z3, normalize=False, tau=1.0, alpha=0.5, grad=True, dtype=np.float64):
    z1 = np.asarray(z1, dtype); z2 = np.asarray(z2, dtype); z3 = np.asarray(z3, dtype)
    raw = (z1, z2, z3)
    if normalize:                                                         # losses.py:180-185
        norms = [np.linalg.norm(z, axis=-1, keepdims=True) for z in raw]
        u1, u2, u3 = [z / nn for z, nn in zip(raw, norms)]
    else:
        u1, u2, u3 = raw
    neg = u1 @ u3.T                                                       # losses.py:187
    pos = (u1 * u2).sum(-1)                                               # losses.py:188
    x = np.concatenate([neg, pos[:, None]], 1) / tau                      # losses.py:190-193
    lse = _lse(x, 1)
    loss_pos = -pos / tau
    loss_i = 2.0 * (alpha * loss_pos + (1.0 - alpha) * lse)               # losses.py:198
    out = dict(loss_mean=loss_i.mean(), loss_i=loss_i, pos_mean=loss_pos.mean(), neg_mean=lse.mean(), lse=lse)
    if not grad:
        return out
    B = z1.shape[0]
    w = np.exp(x - lse[:, None])
    c_lse = 2.0 * (1.0 - alpha) / B
    dneg = c_lse * w[:, :-1] / tau
    dpos = -2.0 * alpha / (B * tau) + c_lse * w[:, -1] / tau
    du1 = dneg @ u3 + dpos[:, None] * u2
    du3 = dneg.T @ u1
    du2 = dpos[:, None] * u1
    if normalize:
        def back(du, u, nn):
            return (du - u * (du * u).sum(-1, keepdims=True)) / nn
        du1, du2, du3 = back(du1, u1, norms[0]), back(du2, u2, norms[1]), back(du3, u3, norms[2])
    out.update(dz1=du1, dz2=du2, dz3=du3)
    return out


# --------------------------------------------------------------------------------------
# MLP encoder (reference: encoders.py:36-85; heads layers.py:48-91)
# --------------------------------------------------------------------------------------
class MLPParams:
    """Weights of get_mlp's nn.Sequential: Linear(+LeakyReLU 0.01) x k, final Linear, optional head."""

    def __init__(self, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray],
                 head: Optional[str] = None, head_param: Optional[np.ndarray] = None,
                 slope: float = 0.01):
        self.W = [np.asarray(w, np.float64) for w in weights]
        self.b = [np.asarray(b, np.float64) for b in biases]
        self.head = head
        self.head_param = None if head_param is None else np.asarray(head_param, np.float64)
        self.slope = slope


def rescale_head(x, r):
    """RescaleLayer mode 'eq': x / ||x||_2 * r  (layers.py:63-66)."""
    nrm = np.linalg.norm(x, axis=-1, keepdims=True)
    return x / nrm * r, nrm


def softclip_head(x, bound):
    """SoftclipLayer: sigmoid(x) * bound  (layers.py:87-91)."""
    s = 1.0 / (1.0 + np.exp(-x))
    return s * bound[None, :], s


def mlp_forward(P: MLPParams, x: np.ndarray):
    """Returns (y, cache).  encoders.py:38-48: Linear -> LeakyReLU on all but the last layer."""
    acts = [np.asarray(x, np.float64)]
    L = len(P.W)
    for l in range(L):
        z = acts[-1] @ P.W[l].T + P.b[l]
        if l < L - 1:
            z = np.where(z > 0, z, P.slope * z)
        acts.append(z)
    y = acts[-1]
    cache = dict(acts=acts)
    if P.head in ("fixed_sphere", "learnable_sphere"):
        y, nrm = rescale_head(y, P.head_param)
        cache["nrm"] = nrm
    elif P.head in ("fixed_box", "learnable_box"):
        y, s = softclip_head(y, P.head_param)
        cache["sig"] = s
    return y, cache


def mlp_backward(P: MLPParams, cache, gy: np.ndarray):
    """Returns dict(dW=[...], db=[...], dx=..., dhead=...)."""
    acts = cache["acts"]
    g = np.asarray(gy, np.float64)
    dhead = None
    if P.head in ("fixed_sphere", "learnable_sphere"):
        pre = acts[-1]; nrm = cache["nrm"]; u = pre / nrm
        dhead = np.asarray([(g * u).sum()])
        g = P.head_param * (g - u * (g * u).sum(-1, keepdims=True)) / nrm
    elif P.head in ("fixed_box", "learnable_box"):
        s = cache["sig"]
        dhead = (g * s).sum(0)
        g = g * P.head_param[None, :] * s * (1.0 - s)
    L = len(P.W)
    dW = [None] * L; db = [None] * L
    for l in reversed(range(L)):
        if l < L - 1:
            g = g * np.where(acts[l + 1] > 0, 1.0, P.slope)
        dW[l] = g.T @ acts[l]
        db[l] = g.sum(0)
        g = g @ P.W[l]
    return dict(dW=dW, db=db, dx=g, dhead=dhead)


# --------------------------------------------------------------------------------------
# Mixing network g (reference: invertible_network_utils.py:87-123) -- forward only, frozen
# --------------------------------------------------------------------------------------
def mixing_forward(Ws: Sequence[np.ndarray], z: np.ndarray, slope: float = 0.2) -> np.ndarray:
    x = np.asarray(z, np.float64)
    for i, W in enumerate(Ws):
        x = x @ np.asarray(W, np.float64).T
        if i < len(Ws) - 1:
            x = np.where(x > 0, x, slope * x)
    return x


# --------------------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults as used at main_mlp.py:312: betas (0.9, 0.999), eps 1e-8)
# --------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """One Adam update, ``step`` is the 1-based step count.  Returns new (p, m, v)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


# --------------------------------------------------------------------------------------
# One unsupervised train step (reference: main_mlp.py:258-285)
# --------------------------------------------------------------------------------------
def train_step(P: MLPParams, gWs, z1, z2, state, p=2, tau=1.0, lr=1e-4):
    """state = dict(step=int, m=[...], v=[...]) over [W0,b0,W1,b1,...,(head)].  Mutates P/state.

    Returns (loss, pos_mean, neg_mean) as floats.
    """
    B = z1.shape[0]
    x = np.concatenate([mixing_forward(gWs, z1), mixing_forward(gWs, z2)], 0)
    y, cache = mlp_forward(P, x)
    a, b = y[:B], y[B:]
    c = np.roll(a, 1, 0)                                            # main_mlp.py:272
    out = lp_simclr_loss(a, b, c, p=p, tau=tau, compat=True)        # main_mlp.py:143-145
    dz1 = out["dz1"] + np.roll(out["dz3"], -1, 0)                   # roll backward
    gy = np.concatenate([dz1, out["dz2"]], 0)
    gr = mlp_backward(P, cache, gy)
    state["step"] += 1
    flat_p: List[np.ndarray] = []
    flat_g: List[np.ndarray] = []
    for l in range(len(P.W)):
        flat_p += [("W", l), ("b", l)]
        flat_g += [gr["dW"][l], gr["db"][l]]
    if P.head in ("learnable_sphere", "learnable_box"):
        flat_p.append(("h", 0)); flat_g.append(gr["dhead"])
    for k, ((kind, l), g) in enumerate(zip(flat_p, flat_g)):
        cur = P.W[l] if kind == "W" else (P.b[l] if kind == "b" else P.head_param)
        newp, state["m"][k], state["v"][k] = adam_step(cur, g, state["m"][k], state["v"][k], state["step"], lr)
        if kind == "W":
            P.W[l] = newp
        elif kind == "b":
            P.b[l] = newp
        else:
            P.head_param = newp
    return float(out["loss_mean"]), float(out["pos_mean"]), float(out["neg_mean"])


def supervised_train_step(P: MLPParams, gWs, z1, state, lr=1e-4, return_grads=False):
    """One SUPERVISED train step (reference: main_mlp.py:258-285 with ``test = True``): z1_rec = f(g(z1)),
    total_loss_value = F.mse_loss(z1_rec, z1) (:274-276; h(z2) is evaluated and unused), backward, Adam.  Mutates P / state
    like train_step.  Returns the loss (and the parameter gradients [dW0, db0, ...] if asked)."""
    z1 = np.asarray(z1, np.float64)
    y, cache = mlp_forward(P, mixing_forward(gWs, z1))
    diff = y - z1
    loss = float((diff * diff).mean())                              # F.mse_loss: mean over all B * n elements
    gr = mlp_backward(P, cache, 2.0 * diff / diff.size)
    state["step"] += 1
    flat = []
    for l in range(len(P.W)):
        flat += [("W", l, gr["dW"][l]), ("b", l, gr["db"][l])]
    if P.head in ("learnable_sphere", "learnable_box"):
        flat.append(("h", 0, gr["dhead"]))
    for k, (kind, l, g) in enumerate(flat):
        cur = P.W[l] if kind == "W" else (P.b[l] if kind == "b" else P.head_param)
        newp, state["m"][k], state["v"][k] = adam_step(cur, g, state["m"][k], state["v"][k], state["step"], lr)
        if kind == "W":
            P.W[l] = newp
        elif kind == "b":
            P.b[l] = newp
        else:
            P.head_param = newp
    return (loss, [g for _, _, g in flat]) if return_grads else loss


def flat_l2_search(table, query, k, chunk=256):
    """Exact k nearest rows in squared L2, ascending, ties to the lower row -- what faiss.IndexFlatL2.search returns at
    /root/reference/datasets/threedident_dataset.py:104-105 (faiss itself is not in the image: exact search is its
    definition; this function is pinned by construction on known-answer grids, tests/test_oracle_golden.py).
    Returns (D (Q, k) float64, I (Q, k) int64)."""
    table = np.asarray(table, np.float64); query = np.asarray(query, np.float64)
    Q = query.shape[0]
    D = np.empty((Q, k)); I = np.empty((Q, k), np.int64)
    for a in range(0, Q, chunk):
        q = query[a:a + chunk]
        d = np.zeros((q.shape[0], table.shape[0]))
        for c in range(table.shape[1]):                    # coordinate by coordinate: no (Q, N, n) temporary
            d += (q[:, c:c + 1] - table[None, :, c]) ** 2
        order = np.argsort(d, axis=1, kind="stable")[:, :k]
        I[a:a + chunk] = order
        D[a:a + chunk] = np.take_along_axis(d, order, 1)
    return D, I


def threedident_snap(table, z, z_tilde):
    """threedident_dataset.py:104-116: nearest grid row of z, nearest of z~ that differs from it."""
    _, iz = flat_l2_search(table, z, 1)
    _, izt = flat_l2_search(table, z_tilde, 2)
    iz = iz[:, 0]
    return iz, np.where(izt[:, 0] != iz, izt[:, 0], izt[:, 1])


# ------------------------------------------------------------------------------------------------ KITTI-masks pair assembly
def kitti_getitem(data, latents, cumlens, index, t_steps_forward):
    """Restatement of KittiMasks.__getitem__ (/root/reference/kitti_masks/dataset.py:90-131, transform=None) with the random time
    step passed in (the reference draws ``np.random.randint(1, max_delta_t + 1)`` at :97).  data / latents: lists of per-sequence
    arrays.  PARITY UNPINNED against the reference's own execution: its module imports torchvision / matplotlib, absent here."""
    sequence_ind = int(np.searchsorted(cumlens, index, side="right"))                       # :91
    start_ind = index if sequence_ind == 0 else index - cumlens[sequence_ind - 1]           # :92-95
    seq_len = len(data[sequence_ind])                                                       # :96
    end_ind = min(start_ind + t_steps_forward, seq_len - 1)                                 # :98
    first = np.asarray(data[sequence_ind][start_ind]).astype(np.uint8) * 255                # :100
    second = np.asarray(data[sequence_ind][end_ind]).astype(np.uint8) * 255                 # :101
    l1, l2 = latents[sequence_ind][start_ind], latents[sequence_ind][end_ind]               # :103-108
    first, second = first[None], second[None]                                               # :123-125 channel dim
    first, second = first.astype(np.float32) / 255.0, second.astype(np.float32) / 255.0     # :127-131
    return first, second, l1, l2


def kitti_collate(sample):
    """custom_collate (/root/reference/kitti_masks/dataset.py:134-142): first / second samples and labels interleaved."""
    inputs, labels = [], []
    for s in sample:
        inputs += [s[0], s[1]]
        labels += [s[2], s[3]]
    return np.stack(inputs), np.stack(labels)
