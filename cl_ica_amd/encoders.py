"""MLP encoder factory with the reference's signature (/root/reference/encoders.py:10-85).

``get_mlp`` returns an ``nn.Sequential`` subclass holding exactly the modules the reference would
build (``nn.Linear`` / ``nn.LeakyReLU`` / head), so ``state_dict()`` keys and shapes
(``0.weight, 0.bias, 2.weight, ...``), ``.parameters()`` order, indexing (``f[-1].r``) and the
default initialisation are identical -- checkpoints load both ways.  ``forward`` does not call the
child modules: the Linear(+bias)(+LeakyReLU) stack runs as fused fp32-MFMA GEMM kernels
(cl_ica_amd/csrc/linear.hip) through one autograd node.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from . import layers as ls
from . import ops

__all__ = ["get_mlp", "FusedMLP"]


class _MLPStackFn(torch.autograd.Function):
    """Linear -> LeakyReLU(slope) -> ... -> Linear (no activation after the last)."""

    @staticmethod
    def forward(ctx, x, slope, *params):
        L = len(params) // 2
        acts = [x.detach()]
        for l in range(L):
            acts.append(ops.linear_fwd(acts[-1], params[2 * l], params[2 * l + 1], leaky=(l < L - 1), slope=slope))
        ctx.slope = slope
        ctx.L = L
        ctx.save_for_backward(*acts[:-1], *[p.detach() for p in params[0::2]])
        return acts[-1]

    @staticmethod
    def backward(ctx, gy):
        L, slope = ctx.L, ctx.slope
        saved = ctx.saved_tensors
        acts, Ws = saved[:L], saved[L:]
        grads: List[Optional[torch.Tensor]] = [None] * (2 * L)
        g = gy
        for l in reversed(range(L)):
            need_w, need_b = ctx.needs_input_grad[2 + 2 * l], ctx.needs_input_grad[3 + 2 * l]
            if need_w or need_b:
                dW, db = ops.linear_wgrad(g, acts[l], want_bias=need_b)
                grads[2 * l] = dW if need_w else None
                grads[2 * l + 1] = db if need_b else None
            if l > 0:
                # acts[l] is the LeakyReLU output feeding layer l: its sign gives act'
                g = ops.linear_dgrad(g, Ws[l], acts[l], slope)
            elif ctx.needs_input_grad[0]:
                g = ops.linear_dgrad(g, Ws[0], None, slope)
            else:
                g = None
        return (g, None, *grads)


class FusedMLP(nn.Sequential):
    """``nn.Sequential`` whose forward runs the fused HIP path (same modules, same state dict)."""

    def forward(self, x):
        mods = list(self)
        linears = [m for m in mods if isinstance(m, nn.Linear)]
        slopes = {m.negative_slope for m in mods if isinstance(m, nn.LeakyReLU)}
        slope = slopes.pop() if slopes else 0.01
        if x.dim() != 2:
            x = x.reshape(-1, x.shape[-1])
        params = []
        for lin in linears:
            params += [lin.weight, lin.bias]
        y = _MLPStackFn.apply(x, slope, *params)
        for m in mods:
            if isinstance(m, (ls.RescaleLayer, ls.SoftclipLayer)):
                y = m(y)
        return y


def get_mlp(n_in: int, n_out: int, layers: List[int], layer_normalization: Optional[str] = None,
            output_normalization: Optional[str] = None, output_normalization_kwargs=None):
    """Creates an MLP (same arguments as encoders.py:10-23).

    Args:
        n_in: dimensionality of the input data
        n_out: dimensionality of the output data
        layers: number of neurons for each hidden layer (the reference appends ``n_out`` to the
            caller's list in place, encoders.py:56; reproduced)
        layer_normalization: must be None -- "bn"/"gn" are never used by the reference's drivers
            and have no fused kernel here
        output_normalization: None | "fixed_sphere" | "learnable_sphere" | "fixed_box" | "learnable_box"
        output_normalization_kwargs: forwarded to the head (e.g. ``init_r`` for the sphere)
    """
    if layer_normalization is not None:
        raise NotImplementedError("layer_normalization (bn/gn) is outside the fused hot path; "
                                  "no main_*.py driver of the reference uses it")
    if len(layers) == 0:
        raise ValueError("get_mlp needs at least one hidden layer (the reference's empty-layers "
                         "branch raises as well, encoders.py:54)")
    modules: List[nn.Module] = []
    layers.append(n_out)
    width = n_in
    for i, l in enumerate(layers):
        modules.append(nn.Linear(width, l))
        if i < len(layers) - 1:
            modules.append(nn.LeakyReLU())
        width = l
    kw = output_normalization_kwargs or {}
    if output_normalization == "fixed_sphere":
        modules.append(ls.RescaleLayer(fixed_r=True, **kw))
    elif output_normalization == "learnable_sphere":
        modules.append(ls.RescaleLayer(init_r=1.0, fixed_r=False))
    elif output_normalization == "fixed_box":
        modules.append(ls.SoftclipLayer(n=n_out, fixed_abs_bound=True, **kw))
    elif output_normalization == "learnable_box":
        modules.append(ls.SoftclipLayer(n=n_out, fixed_abs_bound=False, **kw))
    elif output_normalization is not None:
        raise ValueError("output_normalization")
    return FusedMLP(*modules)
