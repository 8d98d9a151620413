"""Spaces combined with marginal / conditional densities (same surface as
/root/reference/latent_spaces.py:8-75)."""
from __future__ import annotations

from typing import Callable, List

import torch

from .spaces import Space

__all__ = ["LatentSpace", "ProductLatentSpace"]


class LatentSpace:
    """Combines a topological space with a marginal and conditional density to sample from."""

    def __init__(self, space: Space, sample_marginal: Callable, sample_conditional: Callable):
        self.space = space
        self._sample_marginal = sample_marginal
        self._sample_conditional = sample_conditional

    def _bound(self, fn, what):
        if fn is None:
            raise RuntimeError(f"{what} was not set")
        return lambda *args, **kwargs: fn(self.space, *args, **kwargs)

    @property
    def sample_conditional(self):
        return self._bound(self._sample_conditional, "sample_conditional")

    @sample_conditional.setter
    def sample_conditional(self, value: Callable):
        assert callable(value)
        self._sample_conditional = value

    @property
    def sample_marginal(self):
        return self._bound(self._sample_marginal, "sample_marginal")

    @sample_marginal.setter
    def sample_marginal(self, value: Callable):
        assert callable(value)
        self._sample_marginal = value

    @property
    def dim(self):
        return self.space.dim


class ProductLatentSpace(LatentSpace):
    """Cartesian product of latent spaces: samples are concatenated along the feature axis and the
    conditional is applied block-wise (latent_spaces.py:49-75)."""

    def __init__(self, spaces: List[LatentSpace]):
        self.spaces = spaces

    def sample_conditional(self, z, size, **kwargs):
        parts, n = [], 0
        for s in self.spaces:
            parts.append(s.sample_conditional(z=z[..., n:n + s.dim], size=size, **kwargs))
            n += s.dim
        return torch.cat(parts, -1)

    def sample_marginal(self, size, **kwargs):
        return torch.cat([s.sample_marginal(size=size, **kwargs) for s in self.spaces], -1)

    @property
    def dim(self):
        return sum(s.dim for s in self.spaces)
