"""Output-normalisation layers with the reference's constructor/attribute surface
(/root/reference/layers.py:30-91), forward/backward on the HIP head kernels."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import lazy, ops

__all__ = ["Lambda", "Flatten", "RescaleLayer", "SoftclipLayer", "LeakyReLU"]


class Lambda(nn.Module):
    """Module wrapper around an arbitrary callable (reference layers.py:30-38): ``Lambda(f)(*a, **k) == f(*a, **k)``."""

    def __init__(self, f):
        super().__init__()
        self.f = f

    def forward(self, *inputs, **options):
        fn = self.f
        return fn(*inputs, **options)


class Flatten(Lambda):
    """Keeps the batch axis, merges all others (reference layers.py:41-45)."""

    def __init__(self):
        super().__init__(self._merge_trailing_axes)

    @staticmethod
    def _merge_trailing_axes(batch):
        return batch.view(batch.shape[0], -1)


def _scale_tensor(shape, value, learnable: bool):
    """The reference keeps a fixed scale as a plain tensor attribute (neither parameter nor buffer, so it is absent from
    the state dict) and a learnable one as an ``nn.Parameter`` -- checkpoints only interchange if this is reproduced."""
    t = torch.ones(shape, dtype=torch.float32) * value        # `value` may be a number or a broadcastable tensor
    return nn.Parameter(t) if learnable else t.detach()


class _RescaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r):
        y, inv = ops.rescale_fwd(x, r)
        ctx.save_for_backward(x.detach(), r.detach(), inv)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, r, inv = ctx.saved_tensors
        dx, dr = ops.rescale_bwd(x, r, inv, gy, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dr


class _SoftclipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x.detach(), bound.detach())
        return ops.softclip_fwd(x, bound)

    @staticmethod
    def backward(ctx, gy):
        x, bound = ctx.saved_tensors
        dx, db = ops.softclip_bwd(x, bound, gy, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, db


class RescaleLayer(nn.Module):
    """Normalize the data to a hypersphere with fixed/learnable radius: x / ||x|| * r.

    Attribute surface of layers.py:48-61: ``r`` is an ``nn.Parameter`` of shape (1,) when learnable
    (state-dict key ``<idx>.r``) and a plain tensor -- neither parameter nor buffer -- when fixed.
    Only ``mode="eq"`` is implemented; the reference never uses "leq" (SURVEY.md section 8 A8).
    """

    def __init__(self, init_r=1.0, fixed_r=False, mode: Optional[str] = "eq"):
        super().__init__()
        self.fixed_r = fixed_r
        if mode != "eq":
            raise NotImplementedError("RescaleLayer(mode='leq') is unused by the reference and not built")
        self.mode = mode
        self.r = _scale_tensor(1, init_r, learnable=not fixed_r)

    def forward(self, x):
        x = lazy.plain(x)
        r = self.r.to(x.device)
        if not r.is_contiguous():
            r = r.contiguous()
        return _RescaleFn.apply(x, r)


class SoftclipLayer(nn.Module):
    """Normalize the data to a hyperrectangle with fixed/learnable size: sigmoid(x) * bound
    (layers.py:74-91; learnable bound has state-dict key ``<idx>.max_abs_bound``)."""

    def __init__(self, n, init_abs_bound=1.0, fixed_abs_bound=True):
        super().__init__()
        self.fixed_abs_bound = fixed_abs_bound
        self.max_abs_bound = _scale_tensor(n, init_abs_bound, learnable=not fixed_abs_bound)

    def forward(self, x):
        x = lazy.plain(x)
        return _SoftclipFn.apply(x, self.max_abs_bound.to(x.device).contiguous())


class _LeakyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        y = ops.leaky_relu_fwd(x, slope)
        ctx.save_for_backward(y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return ops.leaky_relu_bwd(y, gy, ctx.slope), None


class LeakyReLU(nn.LeakyReLU):
    """``nn.LeakyReLU`` (same constructor, no parameters) on the HIP element-wise kernel: the activation the 3DIdent encoder
    puts between the backbone's output and the head's Linear (main_3dident.py:365-370).  The backward recovers the sign of
    the input from the saved OUTPUT, which needs a strictly positive slope; other slopes (``nn.LeakyReLU`` accepts them) are
    refused at construction instead of failing at the first backward."""

    def __init__(self, negative_slope: float = 0.01, inplace: bool = False):
        if not float(negative_slope) > 0.0:
            raise ValueError(f"cl_ica_amd.layers.LeakyReLU needs negative_slope > 0 (got {negative_slope}); use torch.nn.ReLU / "
                             "torch.nn.LeakyReLU for a zero or negative slope")
        super().__init__(negative_slope, inplace)

    def forward(self, x):
        x = lazy.plain(x)
        if x.dim() != 2:
            return _LeakyFn.apply(x.reshape(-1, x.shape[-1]), float(self.negative_slope)).reshape(x.shape)
        return _LeakyFn.apply(x, float(self.negative_slope))
