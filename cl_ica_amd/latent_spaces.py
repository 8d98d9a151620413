"""Latent spaces = a topological space plus the two densities a contrastive step draws from.

Call surface of /root/reference/latent_spaces.py (LatentSpace :8-46, ProductLatentSpace :49-75):
``LatentSpace(space, sample_marginal, sample_conditional)`` where the two callables take the space as their
first argument (``lambda space, size, device=...: space.uniform(size, device=device)``), are exposed as
``ls.sample_marginal(size=..., device=...)`` / ``ls.sample_conditional(z=..., size=..., device=...)`` with the
space already bound, may be replaced by assignment, and ``ls.dim`` is the space's dimension.  The product space
concatenates its factors' samples along the feature axis and conditions each factor on its own slice.

The draws themselves come from the on-device samplers behind `cl_ica_amd.spaces`.
"""
from __future__ import annotations

import functools
import itertools
from typing import Callable, Optional, Sequence

import torch

from .spaces import Space

__all__ = ["LatentSpace", "ProductLatentSpace"]


class _BoundSampler:
    """Data descriptor: stores a ``fn(space, ...)`` per instance and hands out ``fn`` with the instance's space bound."""

    def __set_name__(self, owner, name):
        self.public = name
        self.slot = "_" + name + "_fn"

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        fn: Optional[Callable] = getattr(obj, self.slot, None)
        if fn is None:
            raise RuntimeError(f"{self.public} was not set")
        return functools.partial(fn, obj.space)

    def __set__(self, obj, fn):
        if fn is not None and not callable(fn):
            raise TypeError(f"{self.public} must be callable, got {type(fn).__name__}")
        setattr(obj, self.slot, fn)


class LatentSpace:
    """One space with its marginal p(z) and conditional p(z~ | z) samplers."""

    sample_marginal = _BoundSampler()
    sample_conditional = _BoundSampler()

    def __init__(self, space: Space, sample_marginal: Optional[Callable], sample_conditional: Optional[Callable]):
        self.space = space
        self.sample_marginal = sample_marginal
        self.sample_conditional = sample_conditional

    @property
    def dim(self) -> int:
        return self.space.dim


class ProductLatentSpace(LatentSpace):
    """Cartesian product of latent spaces (block-wise sampling, feature-axis concatenation)."""

    # plain methods here: the factors already carry their own bound samplers
    sample_marginal = None
    sample_conditional = None

    def __init__(self, spaces: Sequence[LatentSpace]):
        self.spaces = list(spaces)

    def _offsets(self):
        ends = list(itertools.accumulate(s.dim for s in self.spaces))
        return zip(self.spaces, [0] + ends[:-1], ends)

    def sample_marginal(self, size, **kwargs):  # noqa: F811  (replaces the class attribute above)
        return torch.cat([s.sample_marginal(size=size, **kwargs) for s in self.spaces], dim=-1)

    def sample_conditional(self, z, size, **kwargs):  # noqa: F811
        blocks = [s.sample_conditional(z=z[..., lo:hi], size=size, **kwargs) for s, lo, hi in self._offsets()]
        return torch.cat(blocks, dim=-1)

    @property
    def dim(self) -> int:
        return sum(s.dim for s in self.spaces)
