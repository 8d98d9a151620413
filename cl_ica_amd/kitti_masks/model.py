"""``BetaVAE_H`` conv encoder of the KITTI-masks experiment, drop-in for /root/reference/kitti_masks/model.py:28-110.

Same constructor (``z_dim, nc, box_norm``), same ``self.encoder`` ``nn.Sequential`` layout, hence the same state-dict keys
(``encoder.{0,2,4,6,8}.{weight,bias}``, ``encoder.11.{weight,bias}``, ``encoder.12.max_abs_bound`` with ``box_norm``), the
same Kaiming-normal initialisation with zero biases (:76-79, :102-110) and the same ``forward(x) -> (B, z_dim)``.

Execution: on the GPU the five ``Conv2d(k=4) + ReLU`` stages run on the HIP library (implicit-GEMM stages in the f16x2 split
arithmetic since round 5, ``cl_ica_amd/conv.py`` / ``clica_conv16_*`` / ``clica_conv_*``; ``CLICA_CONV=miopen`` selects PyTorch-ROCm / MIOpen
as an explicit A/B switch only); everything behind the ``View`` -- ``Linear(256 -> z_dim)`` forward / dgrad / wgrad
(``clica_linear_*``), the learnable Softclip head (``clica_softclip_*``) -- runs on the HIP kernels, and the result feeds
``cl_ica_amd.losses.LpSimCLRLoss`` through strided ``mu[::2]`` / ``mu[1::2]`` views without a copy.
"""
from __future__ import annotations

import torch
from torch import nn

import os

from .. import layers
from ..conv import conv_stack
from ..encoders import _MLPStackFn

__all__ = ["BetaVAE_H", "View", "kaiming_init"]


class View(nn.Module):
    """Reshape stage between the conv stack and the Linear (a module so that the Sequential indices match the reference's)."""

    def __init__(self, size):
        super().__init__()
        self.size = tuple(size)

    def forward(self, tensor):
        return tensor.view(self.size)


class _HipLinear(nn.Linear):
    """``nn.Linear`` (same parameters / state-dict entries) whose forward and backward are the HIP GEMM kernels."""

    def forward(self, x):
        return _MLPStackFn.apply(x.contiguous(), 0.0, self.weight, self.bias)


# (out_channels, kernel, stride, padding) of the five conv stages, model.py:42-51; spatial size 64 -> 32 -> 16 -> 8 -> 4 -> 1
_CONV_STAGES = ((32, 4, 2, 1), (32, 4, 2, 1), (64, 4, 2, 1), (64, 4, 2, 1), (256, 4, 1, 0))


class BetaVAE_H(nn.Module):
    """Encoder half of the beta-VAE architecture (Higgins et al., ICLR 2017) used as contrastive encoder."""

    def __init__(self, z_dim=10, nc=3, box_norm=False):
        super().__init__()
        self.z_dim, self.nc = z_dim, nc
        stages, width = [], nc
        for out_ch, kernel, stride, pad in _CONV_STAGES:          # indices 0..9: Conv2d, ReLU alternating (parameter holders: the stages run on cl_ica_amd/conv.py)
            stages += [nn.Conv2d(width, out_ch, kernel, stride, pad), nn.ReLU(True)]
            width = out_ch
        head = layers.SoftclipLayer(n=z_dim, init_abs_bound=1.0, fixed_abs_bound=False) if box_norm else layers.Lambda(_identity)
        # index 10: (B, 256, 1, 1) -> (B, 256); 11: Linear on the HIP GEMMs; 12: identity | learnable Softclip (HIP)
        self.encoder = nn.Sequential(*stages, View((-1, width)), _HipLinear(width, z_dim), head)
        self.weight_init()

    def weight_init(self):
        # module order = the reference's loop over self._modules blocks (:76-79): same RNG consumption, same initial weights
        for m in self.encoder:
            kaiming_init(m)

    def _encode(self, x):
        # (an input that requires grad with more than four channels: the input-image gradient kernel covers nc <= 4, such a call keeps
        #  nn.Conv2d as round 4 had it -- ADVICE r5; the reference's masks and the benchmark's images have one channel)
        if x.is_cuda and _hip_convs() and not (x.requires_grad and torch.is_grad_enabled() and x.shape[1] > 4):
            # the five Conv2d + ReLU stages as ONE autograd node on the HIP library (cl_ica_amd/conv.py), input-image gradient included;
            # the Conv2d modules keep the parameters.  CLICA_CONV=miopen runs them through nn.Conv2d instead (A/B switch only).
            feats = conv_stack(x.float(), [self.encoder[i] for i in (0, 2, 4, 6, 8)])
            for stage in self.encoder[10:]:
                feats = stage(feats)
            return feats
        return self.encoder(x)

    def forward(self, x, return_z=False):
        return self._encode(x)


def _identity(x):
    return x


def _hip_convs() -> bool:
    return os.environ.get("CLICA_CONV", "hip") != "miopen"


def kaiming_init(m):
    """model.py:102-110: Kaiming-normal weights and zero biases for Linear / Conv2d, unit scale for batch norms."""
    if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
        nn.init.ones_(m.weight)
    elif isinstance(m, (nn.Linear, nn.Conv2d)):
        nn.init.kaiming_normal_(m.weight)
    else:
        return
    if m.bias is not None:
        nn.init.zeros_(m.bias)
