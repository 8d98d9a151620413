"""3DIdent experiment: encoder head, loss selection and ``train_step`` of /root/reference/main_3dident.py on the HIP path.

What runs where (BASELINE.json configs[3]: "ResNet-18 encoder ... conv path via PyTorch-ROCm + HIP loss"):
  * the backbone (``torchvision.models.resnet*`` in the reference, :287-292) is ANY ``nn.Module`` mapping images to
    ``(B, 10 * n_latents)`` features and stays on PyTorch-ROCm / MIOpen -- torchvision is not part of this image, so
    ``setup_f`` takes the backbone constructor as an argument and only falls back to torchvision when it is importable;
  * everything behind it is HIP: ``LeakyReLU`` (``clica_leaky_relu_*``), ``Linear(10 n_lat -> n_lat)``
    (``clica_linear_*``), the rescaling layer (``clica_rescale_*`` / ``clica_softclip_*``), the losses on column slices
    ``z[:, :k]`` (strided views, no copies) and -- optionally -- Adam (``cl_ica_amd.optim.Adam``).

``setup_f`` reproduces the module layout of :365-371 (``nn.Sequential(backbone, LeakyReLU, Linear, rescaling)``), i.e.
the reference's state-dict keys ``0.*`` (backbone), ``2.weight``, ``2.bias``, ``3.r`` / ``3.max_abs_bound``.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import nn

from . import layers, lazy, losses
from .encoders import _MLPStackFn

__all__ = ["setup_f", "make_unsupervised_loss", "train_step", "HipLinear", "unpack_item_list"]


class HipLinear(nn.Linear):
    """``nn.Linear`` (same parameters / state-dict entries / default init) computed by the HIP GEMM kernels."""

    def forward(self, x):
        x = lazy.plain(x)
        return _MLPStackFn.apply(x if x.is_contiguous() else x.contiguous(), 0.0, self.weight, self.bias)


def _identity(x):
    return x


def make_rescaling(args, n_non_angular_latents: int, n_angular_latents: int) -> nn.Module:
    """The ``rescaling`` module of ``setup_f`` (main_3dident.py:311-345) for the given mode flags."""
    periodic_rescale_layer = layers.RescaleLayer(fixed_r=False, mode="eq")
    if getattr(args, "box_constraint", None) is not None:
        non_periodic_rescale_layer = layers.SoftclipLayer(n=n_non_angular_latents, fixed_abs_bound=args.box_constraint == "fix")
    elif getattr(args, "sphere_constraint", None) is not None:
        non_periodic_rescale_layer = layers.RescaleLayer(fixed_r=args.sphere_constraint == "fix", mode="eq")
    else:
        non_periodic_rescale_layer = layers.Lambda(_identity)
    np_rc = getattr(args, "non_periodic_rotation_and_color", False)
    if getattr(args, "position_only", False):
        return non_periodic_rescale_layer
    if getattr(args, "rotation_and_color_only", False) or getattr(args, "rotation_only", False) or getattr(args, "color_only", False):
        return non_periodic_rescale_layer if np_rc else periodic_rescale_layer
    if np_rc:
        return non_periodic_rescale_layer
    k = n_non_angular_latents
    return layers.Lambda(lambda x: torch.cat((non_periodic_rescale_layer(x[:, :k]), periodic_rescale_layer(x[:, k:])), dim=1))


def setup_f(args, n_non_angular_latents: int, n_angular_latents: int,
            base_encoder: Optional[Callable[..., nn.Module]] = None) -> nn.Module:
    """``nn.Sequential(backbone(num_classes=10 n_lat), LeakyReLU, Linear(10 n_lat, n_lat), rescaling)`` (:365-370).
    ``base_encoder(pretrained, num_classes=...)`` builds the backbone; default: ``torchvision.models.<args.encoder>``."""
    if getattr(args, "identity_solution", False):
        return nn.Sequential(layers.Flatten())
    n_latents = n_non_angular_latents + n_angular_latents
    rescaling = make_rescaling(args, n_non_angular_latents, n_angular_latents)
    if base_encoder is None:
        try:
            from torchvision import models       # not in this image; present in the reference's environment
            base_encoder = {"rn18": models.resnet18, "rn50": models.resnet50, "rn101": models.resnet101,
                            "rn152": models.resnet152}[args.encoder]
        except ImportError as e:
            if getattr(args, "encoder", "rn18") != "rn18":
                raise ImportError("torchvision is not installed: pass base_encoder=<callable building the backbone> "
                                  "(only the default rn18 has a stand-in, cl_ica_amd/resnet.py)") from e
            from .resnet import resnet18 as base_encoder      # same architecture and state-dict keys as torchvision's
    return nn.Sequential(base_encoder(False, num_classes=n_latents * 10), layers.LeakyReLU(),
                         HipLinear(n_latents * 10, n_latents), rescaling)


def make_unsupervised_loss(args, n_non_angular_latents: int):
    """Loss selection of ``train_unsupervised`` (main_3dident.py:402-445)."""
    spherical_loss = losses.SimCLRLoss(normalize=False, tau=1.0)
    kind = getattr(args, "unsupervised_loss", "l2")
    if kind in ("l1", "l2", "l3"):
        nonspherical_loss = losses.LpSimCLRLoss(p=int(kind[1]), tau=1.0, simclr_compatibility_mode=True, pow=True)
    elif kind == "vmf":
        nonspherical_loss = losses.SimCLRLoss(normalize=True, tau=1.0)
    else:
        raise ValueError(f"unsupervised_loss {kind!r}")
    k = n_non_angular_latents

    def loss(z1, z2_con_z1, z3, z1_rec, z2_con_z1_rec, z3_rec):
        # the combined objective; the reference slices z3_rec at a hard-coded 3 (:431,439)
        nsl = nonspherical_loss(z1, z2_con_z1, z3, z1_rec[:, :k], z2_con_z1_rec[:, :k], z3_rec[:, :3])
        sl = spherical_loss(z1, z2_con_z1, z3, z1_rec[:, k:], z2_con_z1_rec[:, k:], z3_rec[:, 3:])
        return sl[0] + nsl[0], [(sl[0], sl[1])] + [(nsl[0], nsl[1])]

    if getattr(args, "position_only", False):
        loss = nonspherical_loss
    elif getattr(args, "rotation_and_color_only", False) or getattr(args, "rotation_only", False) or getattr(args, "color_only", False):
        loss = spherical_loss
    if getattr(args, "non_periodic_rotation_and_color", False):
        loss = nonspherical_loss
    return loss


def unpack_item_list(lst):
    if isinstance(lst, tuple):
        lst = list(lst)
    return [unpack_item_list(it) if isinstance(it, (tuple, list)) else it.item() for it in lst]


def train_step(data, loss, optimizer, f, sync: bool = True):
    """One unsupervised step (main_3dident.py:467-503): two encoder passes, ``z3_rec = roll(z1_rec)`` inside the graph,
    the loss with ``None`` latents, backward, optimizer step.  Returns ``(total, per_item, [pos_mean, neg_mean])`` as python
    floats like the reference when ``sync`` (3 host syncs), device tensors otherwise."""
    (_z1, _z2), (x1, x2_con_x1) = data
    optimizer.zero_grad()
    z1_rec = f(x1)
    z2_con_z1_rec = f(x2_con_x1)
    del x1, x2_con_x1
    z3_rec = torch.roll(z1_rec, 1, 0)
    total_loss_value, total_loss_per_item_value, losses_value = loss(None, None, None, z1_rec, z2_con_z1_rec, z3_rec)
    total_loss_value.backward()
    if hasattr(optimizer, "all_reduce_grads"):
        optimizer.all_reduce_grads()
    optimizer.step()
    if not sync:
        return total_loss_value, total_loss_per_item_value, losses_value
    return total_loss_value.item(), total_loss_per_item_value, unpack_item_list(losses_value)
