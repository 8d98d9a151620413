"""KITTI-masks temporal pairs on the device: counterpart of /root/reference/kitti_masks/dataset.py (``KittiMasks`` :11-131,
``custom_collate`` :134-142, ``return_data`` :145-175).

The reference keeps the pedestrian sequences as a pickled list of bool arrays on the host; every ``__getitem__`` picks frame
``start`` of a sequence and a frame ``1 .. max_delta_t`` steps later (clamped to the sequence's end, :90-98), converts both to
float32, and DataLoader workers interleave ``batch_size // 2`` such pairs into a ``(batch_size, 1, 64, 64)`` batch that is then copied
to the GPU.  Here all frames live in HBM once (uint8, 4 KB per 64 x 64 mask), the index arithmetic of ``__getitem__`` runs on
device tensors for a whole batch, and ONE launch (``clica_kitti_gather_pairs``, csrc/kitti_pairs.hip) writes the interleaved batch
and labels -- no workers, no H2D copy per step.  What is kept: ``KittiMasks(path, transform, max_delta_t)`` with ``__len__`` /
``__getitem__`` (host arrays, like the reference, for the evaluation code that calls them), ``custom_collate``, and
``return_data(args)`` -> an iterable of ``(images, labels)`` batches with the reference's semantics (``batch_size`` halved for
the pairs, shuffled without replacement every epoch, ``drop_last``).

Out of scope: the ``transform="default"`` augmentation (PIL / torchvision RandomAffine, never enabled by ``return_data``) and the
download of ``kitti_peds_v2.pickle`` (no network): pass ``data=`` (the unpickled dict) or put the file under ``path``.
Parity note: the reference module cannot be imported in this image (torchvision, matplotlib): the oracle restates it
(oracle/np_oracle.py: kitti_getitem / kitti_collate) -- parity unpinned against the reference's own execution for this row.
"""
from __future__ import annotations

import os
import pickle
from typing import Optional

import numpy as np
import torch

from .. import ops

__all__ = ["KittiMasks", "custom_collate", "return_data", "DevicePairLoader"]


class KittiMasks:
    """latents encode: 0 centre of mass vertical position, 1 centre of mass horizontal position, 2 area (dataset.py:12-17)."""

    def __init__(self, path="./data/kitti/", transform=None, max_delta_t=5, data: Optional[dict] = None, device="cuda"):
        if transform is not None:
            raise NotImplementedError("KittiMasks(transform=...): the PIL / torchvision augmentation is not built (return_data never enables it)")
        self.path, self.transform, self.max_delta_t = path, None, int(max_delta_t)
        self.fname = "kitti_peds_v2.pickle"
        self.device = torch.device(device)
        if data is None:
            file_path = os.path.join(self.path, self.fname)
            if not os.path.exists(file_path):
                raise FileNotFoundError(f"{file_path} not found (no network here: the reference would download it from zenodo record 3931823)")
            with open(file_path, "rb") as fh:
                data = pickle.load(fh)
        self.data = data["pedestrians"]
        self.latents = data["pedestrians_latents"]
        self.lens = [len(seq) - 1 for seq in self.data]           # the last image of a sequence can never be a starting point (:64-66)
        self.cumlens = np.cumsum(self.lens)
        # device tables: every frame once, sequence s at [seq_start[s], seq_start[s] + len(seq))
        seq_len = np.array([len(seq) for seq in self.data], np.int64)
        self._seq_start_h = np.concatenate([[0], np.cumsum(seq_len)[:-1]]).astype(np.int64)
        frames = np.concatenate([np.asarray(seq).astype(np.uint8) for seq in self.data], 0)
        self.frames = torch.as_tensor(frames, device=self.device).contiguous()
        self.frame_latents = torch.as_tensor(np.concatenate([np.asarray(l, np.float32) for l in self.latents], 0), device=self.device)
        self.seq_start = torch.as_tensor(self._seq_start_h, device=self.device)
        self.seq_len = torch.as_tensor(seq_len, device=self.device)
        self.cumlens_dev = torch.as_tensor(self.cumlens.astype(np.int64), device=self.device)

    def __len__(self):
        return int(self.cumlens[-1])

    # ------------------------------------------------------------------ the index arithmetic of __getitem__ (:91-98), batched
    def pair_frames(self, index: torch.Tensor, t_steps_forward: torch.Tensor):
        """Global frame ids (first, second) of the pairs starting at data-set indices `index` with `t_steps_forward` frames between."""
        index = index.to(self.device, torch.int64)
        t = t_steps_forward.to(self.device, torch.int64)
        seq = torch.searchsorted(self.cumlens_dev, index, right=True)
        prev = torch.where(seq > 0, self.cumlens_dev[(seq - 1).clamp_min(0)], torch.zeros_like(index))
        start = index - prev
        end = torch.minimum(start + t, self.seq_len[seq] - 1)
        base = self.seq_start[seq]
        return base + start, base + end

    def batch(self, index: torch.Tensor, t_steps_forward: torch.Tensor):
        """custom_collate([self[i] for i in index]) with the given time steps: ``(images (2B, 1, H, W), labels (2B, 3))`` on the device."""
        first, second = self.pair_frames(index, t_steps_forward)
        return ops.kitti_gather_pairs(self.frames, first, second, self.frame_latents)

    def sample_time_steps(self, count: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """np.random.randint(1, max_delta_t + 1) per item (:97), drawn on the device."""
        return torch.randint(1, self.max_delta_t + 1, (count,), device=self.device, generator=generator)

    # ------------------------------------------------------------------ host-side item access, as the reference's (evaluation code uses it)
    def __getitem__(self, index):
        t = int(np.random.randint(1, self.max_delta_t + 1))
        f, s = self.pair_frames(torch.tensor([int(index)]), torch.tensor([t]))
        img, lab = ops.kitti_gather_pairs(self.frames, f, s, self.frame_latents)
        img, lab = img.cpu().numpy(), lab.cpu().numpy()
        return img[0], img[1], lab[0], lab[1]

    def sample_observations(self, num, random_state, return_latents=False):
        """Sample a batch of observations X (:68-82; needed by disentanglement_lib-style evaluation): the FIRST frames of `num` items."""
        assert not (num % 2)
        indices = random_state.choice(len(self), num, replace=False)
        t = self.sample_time_steps(num)
        first, _ = self.pair_frames(torch.as_tensor(indices), t)
        img, lab = ops.kitti_gather_pairs(self.frames, first, first, self.frame_latents)
        batch, latents = img[::2].cpu().numpy(), lab[::2].cpu().numpy()
        return (batch, latents) if return_latents else batch

    def sample(self, num, random_state):
        x, y = self.sample_observations(num, random_state, return_latents=True)
        return y, x


def custom_collate(sample):
    """dataset.py:134-142: ``sample`` = list of (first, second, latents1, latents2) -> (inputs, labels) with the pairs interleaved."""
    inputs, labels = [], []
    for s in sample:
        inputs.append(s[0]); inputs.append(s[1])
        labels.append(s[2]); labels.append(s[3])
    return torch.tensor(np.stack(inputs)), torch.tensor(np.stack(labels))


class DevicePairLoader:
    """What ``return_data`` hands the Solver in place of the reference's ``DataLoader(shuffle=True, drop_last=True,
    collate_fn=custom_collate)``: iterating it yields ``len(self)`` batches of ``pairs_per_batch`` pairs per epoch, every start index
    at most once per epoch (a fresh device permutation each epoch), time steps drawn per item."""

    def __init__(self, dataset: KittiMasks, pairs_per_batch: int, shuffle: bool = True, drop_last: bool = True, seed: Optional[int] = None):
        self.dataset, self.pairs, self.shuffle, self.drop_last = dataset, int(pairs_per_batch), shuffle, drop_last
        self.generator = torch.Generator(device=dataset.device)
        if seed is not None:
            self.generator.manual_seed(int(seed))
        self.batch_size = self.pairs

    def __len__(self):
        n = len(self.dataset)
        return n // self.pairs if self.drop_last else (n + self.pairs - 1) // self.pairs

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, device=self.dataset.device, generator=self.generator) if self.shuffle else torch.arange(n, device=self.dataset.device)
        for b in range(len(self)):
            idx = order[b * self.pairs:(b + 1) * self.pairs]
            yield self.dataset.batch(idx, self.dataset.sample_time_steps(idx.numel(), self.generator))


def return_data(args, data: Optional[dict] = None):
    """dataset.py:145-175: ``args.batch_size`` counts IMAGES; the loader works on ``batch_size // 2`` pairs."""
    assert args.image_size == 64, "currently only image size of 64 is supported"
    assert not (args.batch_size % 2)
    if args.dataset.lower() != "kittimasks":
        raise NotImplementedError
    train_data = KittiMasks(max_delta_t=args.kitti_max_delta_t, transform=None, data=data,
                            **({"path": args.dset_dir} if getattr(args, "dset_dir", None) else {}))
    return DevicePairLoader(train_data, args.batch_size // 2, shuffle=True, drop_last=True, seed=getattr(args, "seed", None))
