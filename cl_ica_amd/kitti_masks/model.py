"""``BetaVAE_H`` conv encoder of the KITTI-masks experiment, drop-in for /root/reference/kitti_masks/model.py:28-110.

Same constructor (``z_dim, nc, box_norm``), same ``self.encoder`` ``nn.Sequential`` layout, hence the same state-dict keys
(``encoder.{0,2,4,6,8}.{weight,bias}``, ``encoder.11.{weight,bias}``, ``encoder.12.max_abs_bound`` with ``box_norm``), the
same Kaiming-normal initialisation with zero biases (:76-79, :102-110) and the same ``forward(x) -> (B, z_dim)``.

Execution: the five ``Conv2d(k=4) + ReLU`` stages run on PyTorch-ROCm / MIOpen (north_star keeps the conv path there,
BASELINE.json configs[4]); everything behind the ``View`` -- ``Linear(256 -> z_dim)`` forward / dgrad / wgrad
(``clica_linear_*``), the learnable Softclip head (``clica_softclip_*``) -- runs on the HIP kernels, and the result feeds
``cl_ica_amd.losses.LpSimCLRLoss`` through strided ``mu[::2]`` / ``mu[1::2]`` views without a copy.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import layers
from ..encoders import _MLPStackFn

__all__ = ["BetaVAE_H", "View", "kaiming_init"]


class View(nn.Module):
    def __init__(self, size):
        super().__init__()
        self.size = size

    def forward(self, tensor):
        return tensor.view(self.size)


class _HipLinear(nn.Linear):
    """``nn.Linear`` (same parameters / state-dict entries) whose forward and backward are the HIP GEMM kernels."""

    def forward(self, x):
        return _MLPStackFn.apply(x.contiguous(), 0.0, self.weight, self.bias)


class BetaVAE_H(nn.Module):
    """Encoder half of the beta-VAE architecture (Higgins et al., ICLR 2017) used as contrastive encoder."""

    def __init__(self, z_dim=10, nc=3, box_norm=False):
        super().__init__()
        self.z_dim = z_dim
        self.nc = nc
        if box_norm:
            non_periodic_rescale_layer = layers.SoftclipLayer(n=z_dim, init_abs_bound=1.0, fixed_abs_bound=False)
        else:
            non_periodic_rescale_layer = layers.Lambda(_identity)
        self.encoder = nn.Sequential(
            nn.Conv2d(nc, 32, 4, 2, 1), nn.ReLU(True),        # B,  32, 32, 32
            nn.Conv2d(32, 32, 4, 2, 1), nn.ReLU(True),        # B,  32, 16, 16
            nn.Conv2d(32, 64, 4, 2, 1), nn.ReLU(True),        # B,  64,  8,  8
            nn.Conv2d(64, 64, 4, 2, 1), nn.ReLU(True),        # B,  64,  4,  4
            nn.Conv2d(64, 256, 4, 1), nn.ReLU(True),          # B, 256,  1,  1
            View((-1, 256 * 1 * 1)),                          # B, 256
            _HipLinear(256, z_dim),                           # B, z_dim        (HIP)
            non_periodic_rescale_layer,                       # identity | learnable Softclip (HIP)
        )
        self.weight_init()

    def weight_init(self):
        for block in self._modules:
            for m in self._modules[block]:
                kaiming_init(m)

    def forward(self, x, return_z=False):
        return self._encode(x)

    def _encode(self, x):
        return self.encoder(x)


def _identity(x):
    return x


def kaiming_init(m):
    if isinstance(m, (nn.Linear, nn.Conv2d)):
        nn.init.kaiming_normal_(m.weight)
        if m.bias is not None:
            m.bias.data.fill_(0)
    elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
        m.weight.data.fill_(1)
        if m.bias is not None:
            m.bias.data.fill_(0)
