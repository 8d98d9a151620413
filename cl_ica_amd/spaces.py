"""Latent spaces with on-device samplers: same class/method surface as /root/reference/spaces.py
(``NRealSpace`` / ``NSphereSpace`` / ``NBoxSpace`` x ``uniform / normal / laplace /
generalized_normal / von_mises_fisher``), but every draw is one Philox kernel launch on the GPU --
no torch.distributions on the host, no NumPy vMF + H2D copy, no ``.item()`` per rejection round.

Each call consumes a fresh counter ("draw id") of the module-level generator, so successive calls
are independent; ``manual_seed`` resets it.  RNG streams differ from torch's, parity with the
reference is distributional (tests/test_gpu_samplers.py against tests/golden/g9_samplers.npz).
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch

from . import ops

__all__ = ["Space", "NRealSpace", "NSphereSpace", "NBoxSpace", "manual_seed"]

_state = {"seed": 0, "draw": 0}


def manual_seed(seed: int) -> None:
    _state["seed"] = int(seed)
    _state["draw"] = 0


def _draw(space, dist, n, size, device, **kw):
    if device is None or torch.device(device).type != "cuda":
        raise RuntimeError(f"cl_ica_amd.spaces samples on the GPU only (device={device!r}); there is no host fallback")
    _state["draw"] += 1
    return ops.sample(space, dist, n, size, torch.device(device), seed=_state["seed"], stream_id=_state["draw"], **kw)


def _mean2d(mean, n, size, device):
    mean = torch.as_tensor(mean, dtype=torch.float32)
    if mean.dim() == 1:
        mean = mean.unsqueeze(0)
    assert mean.dim() == 2 and mean.shape[-1] == n and mean.shape[0] in (1, size)
    return mean.to(device)


class Space(ABC):
    @abstractmethod
    def uniform(self, size, device):
        ...

    @abstractmethod
    def normal(self, mean, std, size, device):
        ...

    @abstractmethod
    def laplace(self, mean, lbd, size, device):
        ...

    @abstractmethod
    def generalized_normal(self, mean, lbd, p, size, device):
        ...

    @property
    @abstractmethod
    def dim(self):
        ...


class _Base(Space):
    kind = "real"
    box = (0.0, 1.0)

    def __init__(self, n):
        self.n = n

    @property
    def dim(self):
        return self.n

    def _cond(self, dist, mean, scale, size, device, shape_p=2.0):
        if torch.is_tensor(scale) and scale.numel() > 1:
            # spaces.py:60-72, 157-166, 297: `std` of normal() may be a (n,), (1, n) or (size, n) tensor (laplace and
            # generalized_normal assert a float there, spaces.py:87, 112; accepted here for all three location-scale kinds)
            return _draw(self.kind, dist, self.n, size, device, mean=_mean2d(mean, self.n, size, device), scale=1.0,
                         shape_p=float(shape_p), box=self.box, scale_vec=scale.to(device=device, dtype=torch.float32))
        return _draw(self.kind, dist, self.n, size, device, mean=_mean2d(mean, self.n, size, device), scale=float(scale),
                     shape_p=float(shape_p), box=self.box)

    def normal(self, mean, std, size, device="cuda"):
        return self._cond("normal", mean, std, size, device)

    def laplace(self, mean, lbd, size, device="cuda"):
        return self._cond("laplace", mean, lbd, size, device)

    def generalized_normal(self, mean, lbd, p, size, device="cuda"):
        return self._cond("gennorm", mean, lbd, size, device, shape_p=p)


class NRealSpace(_Base):
    """Unconstrained space R^N (spaces.py:35-119)."""
    kind = "real"

    def uniform(self, size, device="cuda"):
        raise NotImplementedError("Not defined on R^n")


class NSphereSpace(_Base):
    """Unit hypersphere (spaces.py:122-257); like the reference, ``r`` is stored but samples are unit norm."""
    kind = "sphere"

    def __init__(self, n, r=1):
        super().__init__(n)
        self._n_sub = n - 1
        self.r = r

    def uniform(self, size, device="cuda"):
        return _draw("sphere", "uniform", self.n, size, device)

    def von_mises_fisher(self, mean, kappa, size, device="cuda"):
        return self._cond("vmf", mean, kappa, size, device)


class NBoxSpace(_Base):
    """Box [min_, max_]^N (spaces.py:260-351); conditionals are truncated per element."""
    kind = "box"

    def __init__(self, n, min_=-1, max_=1):
        super().__init__(n)
        self.min_, self.max_ = min_, max_
        self.box = (float(min_), float(max_))

    def uniform(self, size, device="cuda"):
        return _draw("box", "uniform", self.n, size, device, box=self.box)
