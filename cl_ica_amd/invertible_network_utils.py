"""Mixing network g (host-side construction, device-side forward).

Counterpart of /root/reference/invertible_network_utils.py:15-123.  Construction is a one-off host
cost and stays in NumPy on purpose: it consumes ``np.random`` exactly like the reference (a pool of
column-normalised U(-1,1) matrices ranked by condition number, then per-layer rejection sampling),
so ``np.random.seed(s)`` reproduces the reference's weights bit-for-bit (KAT: seed 0, n=10, L=3 ->
threshold 4.085284; tests/test_host_logic.py).  The forward runs in one HIP kernel
(csrc/mixing.hip) -- g is frozen, so there is no backward.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import ops

__all__ = ["construct_invertible_mlp", "MixingMLP"]

# act_fct -> (kernel activation kind, its parameter, the nn.Module the reference puts between the Linear layers)
class SmoothLeakyReLU(nn.Module):
    """alpha x + (1 - alpha) log(1 + e^x) (reference invertible_network_utils.py:43-49); parameter-free."""

    def __init__(self, alpha=0.2):
        super().__init__()
        self.alpha = alpha

    def forward(self, x):
        return self.alpha * x + (1 - self.alpha) * torch.log(1 + torch.exp(x))


_ACTS = {"leaky_relu": (0, 0.2, lambda: nn.LeakyReLU(negative_slope=0.2)), "relu": (0, 0.0, lambda: nn.ReLU()),
         "elu": (1, 1.0, lambda: nn.ELU(alpha=1.0)), "smooth_leaky_relu": (2, 0.2, lambda: SmoothLeakyReLU(alpha=0.2)),
         "softplus": (3, 1.0, lambda: nn.Softplus(beta=1))}


def _column_normalised(n: int) -> np.ndarray:
    a = np.random.uniform(-1, 1, (n, n))
    return a / np.sqrt((a * a).sum(0))


def mixing_weights(n: int, n_layers: int, n_iter_cond_thresh: int, cond_thresh_ratio: float,
                   weight_matrix_init: str = "pcl", verbose: bool = True):
    """NumPy weights of the mixing MLP, same RNG consumption as the reference (:71-102)."""
    if weight_matrix_init == "rvs":
        from scipy.stats import ortho_group
        return [ortho_group.rvs(n).astype(np.float32) for _ in range(n_layers)], 0.0
    if weight_matrix_init != "pcl":
        raise ValueError(f"weight matrix {weight_matrix_init} not implemented")
    conds = np.sort([np.linalg.cond(_column_normalised(n)) for _ in range(n_iter_cond_thresh)])
    thresh = conds[int(n_iter_cond_thresh * cond_thresh_ratio)]
    if verbose:
        print("condition number threshold: {0:f}".format(thresh))
    mats = []
    for i in range(n_layers):
        while True:
            w = _column_normalised(n)
            if np.linalg.cond(w) <= thresh:
                break
        if verbose:
            print(f"layer {i + 1}/{n_layers},  condition number: {np.linalg.cond(w)}")
        mats.append(w.astype(np.float32))
    return mats, float(thresh)


class MixingMLP(nn.Sequential):
    """``nn.Sequential(Linear(n,n,bias=False), LeakyReLU(0.2), ..., Linear)`` with frozen parameters
    (same state-dict keys ``0.weight, 2.weight, ...`` as the reference's g.pth) whose forward is the
    single fused HIP kernel."""

    def __init__(self, mats, slope: float = 0.2, act_fct: str = None):
        if act_fct is None:
            act_fct = "leaky_relu" if slope > 0 else "relu"
        kind, param, make = _ACTS[act_fct]
        if act_fct == "leaky_relu":
            param, make = slope, (lambda: nn.LeakyReLU(negative_slope=slope))
        mods = []
        for i, w in enumerate(mats):
            lin = nn.Linear(w.shape[1], w.shape[0], bias=False)
            lin.weight.data = torch.tensor(w, dtype=torch.float32)
            mods.append(lin)
            if i < len(mats) - 1:
                mods.append(make())
        super().__init__(*mods)
        self.act_fct, self.act_kind = act_fct, kind
        self.slope = param          # the activation's parameter (negative slope / alpha / beta)
        for p in self.parameters():
            p.requires_grad = False
        self._stack = None
        self._stack_key = None

    def weight_stack(self) -> torch.Tensor:
        """Contiguous [L, n, n] copy of the layer weights for the fused kernel.  Rebuilt whenever a weight was replaced or
        written in place (``load_state_dict`` of a reference g.pth, ``.to()``): the key is every weight's storage address and
        version counter."""
        ws = [m.weight for m in self if isinstance(m, nn.Linear)]
        key = tuple((w.data_ptr(), w._version, str(w.device)) for w in ws)
        if self._stack is None or self._stack_key != key:
            self._stack = torch.stack([w.detach() for w in ws]).contiguous()
            self._stack_key = key
        return self._stack

    def _apply(self, fn, *a, **k):   # .to(device) invalidates the cached stack
        self._stack = None
        return super()._apply(fn, *a, **k)

    def forward(self, z):
        return ops.mixing_fwd(z, self.weight_stack(), self.slope, act_kind=self.act_kind)


def construct_invertible_mlp(n: int = 20, n_layers: int = 2, n_iter_cond_thresh: int = 10000,
                             cond_thresh_ratio: float = 0.25, weight_matrix_init: str = "pcl",
                             act_fct: str = "leaky_relu"):
    """Create an (approximately) invertible mixing network based on an MLP (same arguments as the
    reference: "relu", "leaky_relu", "elu", "smooth_leaky_relu", "softplus"; "max_out" raises there as well)."""
    if act_fct == "max_out":
        raise NotImplementedError("max_out is not implemented by the reference either (invertible_network_utils.py:58-59)")
    if act_fct not in _ACTS:
        raise Exception(f"activation function {act_fct} not defined.")
    mats, _ = mixing_weights(n, n_layers, n_iter_cond_thresh, cond_thresh_ratio, weight_matrix_init)
    return MixingMLP(mats, act_fct=act_fct)
