"""The latent lookup of ``ThreeDIdentDataset`` (/root/reference/datasets/threedident_dataset.py:64-127) on the device.

The reference samples ONE latent pair per ``__getitem__`` on the host (``latent_space.sample_marginal(size=1, device="cpu")``),
asks a ``faiss.IndexFlatL2`` built over ``raw_latents.npy`` for the closest rendered grid point (``search(z, 1)``) and the
two closest to the positive (``search(z_tilde, 2)``, taking the second when the first coincides with z's), and loads the
two PNGs.  Here a whole batch is drawn by the on-device samplers and snapped to the table by ONE exact brute-force
squared-L2 kernel (``clica_nn_search``, csrc/nn_search.hip) over the table resident in HBM -- 250 000 x 10 floats = 10 MB.

faiss is a third-party dependency of the reference that is not in this image (the reference pins no version;
``IndexFlatL2`` is exact search by definition: D[i, c] = squared L2 distance to the c-th closest stored vector, ascending,
I = its position in ``add`` order); ``IndexFlatL2`` below has the three members the reference uses -- ``add``, ``search``
and ``ntotal`` -- with device tensors instead of NumPy arrays.  The image side of the dataset (PNG decoding, torchvision
transforms) is out of scope: ``ThreeDIdentLatentPairs`` returns the row indices a loader would read.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np
import torch

from .. import ops

__all__ = ["IndexFlatL2", "ThreeDIdentLatentPairs"]


class IndexFlatL2:
    """``faiss.IndexFlatL2(d)`` as used at threedident_dataset.py:71, 83, 104-105: exact squared-L2 search."""

    def __init__(self, d: int, device="cuda"):
        if not 1 <= int(d) <= 64:
            raise ValueError(f"IndexFlatL2: d={d} must be in 1..64")
        self.d = int(d)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("IndexFlatL2 searches on the GPU only; there is no host fallback")
        self._table = torch.empty((0, self.d), dtype=torch.float32, device=self.device)

    @property
    def ntotal(self) -> int:
        return self._table.shape[0]

    def add(self, x) -> None:
        x = torch.as_tensor(np.asarray(x, dtype=np.float32) if not torch.is_tensor(x) else x, dtype=torch.float32)
        if x.dim() != 2 or x.shape[1] != self.d:
            raise ValueError(f"IndexFlatL2.add: expected (N, {self.d}), got {tuple(x.shape)}")
        self._table = torch.cat([self._table, x.to(self.device)], 0).contiguous()

    def search(self, x, k: int):
        """-> (D, I): (Q, k) float32 squared distances ascending, (Q, k) int64 positions (device tensors)."""
        if self.ntotal == 0:
            raise RuntimeError("IndexFlatL2.search: the index is empty")
        x = torch.as_tensor(np.asarray(x, dtype=np.float32) if not torch.is_tensor(x) else x, dtype=torch.float32).to(self.device)
        if x.dim() == 1:
            x = x.unsqueeze(0)
        return ops.nn_search(self._table, x, int(k))


class ThreeDIdentLatentPairs:
    """Batched ``ThreeDIdentDataset.__getitem__`` without the images (threedident_dataset.py:96-127).

    Args:
        latents: ``raw_latents.npy`` contents (N, d) or the dataset root holding that file (threedident_dataset.py:40).
        latent_space: space with ``sample_marginal`` / ``sample_conditional`` (cl_ica_amd.latent_spaces.LatentSpace).
        latent_dimensions_to_use: optional column subset (threedident_dataset.py:43-46).
    """

    def __init__(self, latents, latent_space, latent_dimensions_to_use: Optional[Sequence[int]] = None, device="cuda"):
        if isinstance(latents, (str, os.PathLike)):
            latents = np.load(os.path.join(latents, "raw_latents.npy"))
        latents = np.asarray(latents)
        self.unfiltered_latents = latents
        if latent_dimensions_to_use is not None:
            latents = np.ascontiguousarray(latents[:, list(latent_dimensions_to_use)])
        if latents.ndim != 2 or latents.shape[0] < 2:
            raise ValueError("ThreeDIdentLatentPairs needs a table of at least two latents (z and z~ are snapped to DIFFERENT rows)")
        if latents.shape[1] != latent_space.dim:
            raise AssertionError(f"Shapes do not match, i.e. {latent_space.dim} vs. {latents.shape}")   # as :50-52
        self.latent_space = latent_space
        self.device = torch.device(device)
        self._index = IndexFlatL2(latents.shape[1], device=self.device)
        self._index.add(latents)
        self.latents = self._index._table            # (N, d) float32 on the device

    def __len__(self) -> int:
        return self._index.ntotal

    def sample(self, batch_size: int):
        """-> (index_z, index_z_tilde, z, z_tilde): grid rows (int64) and their latents for `batch_size` pairs."""
        z = self.latent_space.sample_marginal(size=batch_size, device=self.device)
        z_tilde = self.latent_space.sample_conditional(z, size=batch_size, device=self.device)
        return self.snap(z, z_tilde)

    def snap(self, z: torch.Tensor, z_tilde: torch.Tensor):
        """Closest grid point of z; closest of z~ that is not the same grid point (threedident_dataset.py:104-116)."""
        _, iz = ops.nn_search(self.latents, z, 1, want_dist=False)
        _, izt = ops.nn_search(self.latents, z_tilde, 2, want_dist=False)
        index_z = iz[:, 0]
        index_zt = torch.where(izt[:, 0] != index_z, izt[:, 0], izt[:, 1])
        return index_z, index_zt, self.latents[index_z], self.latents[index_zt]
