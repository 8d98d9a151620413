"""KITTI-masks training driver on the HIP loss / head / optimizer: the counterpart of ``Solver`` in
/root/reference/kitti_masks/solver.py (constructor :20-50, ``train`` :52-96, checkpoints :98-128).

What is kept is the call surface a user of the reference touches: ``Solver(args, data_loader)`` with the same ``args``
fields and public attributes, ``train()`` (returns ``False``, writes ``log.csv`` and the ``last`` / 50 000-iteration
checkpoints in the ``{iter, model_states, optim_states}`` layout), ``save_checkpoint`` / ``load_checkpoint`` / ``net_mode``.
The per-batch arithmetic (:61-74) lives in ``train_iteration``: ``mu = net(x)``, the strided views ``mu[::2]`` /
``mu[1::2]`` go to ``LpSimCLRLoss(p, tau=1, compat)`` without copies, negatives are the rolled first views, then
``zero_grad / backward / step`` on the flat-arena Adam (one launch).  The encoder's convolutions are the hand-written stack of
``cl_ica_amd/conv.py`` (``BetaVAE_H``); ``capture`` turns the whole iteration into one HIP graph.

Data parallel (BASELINE.json configs[4]: "DDP over 4 x MI355X"; the reference itself has none): with an initialised
``torch.distributed`` group of world > 1 every rank encodes its own batch, the negatives pool is the autograd-aware
all-gather of all ranks' first views (``distributed.gather_negatives``: all-gather forward, reduce-scatter backward) and the
flat gradient arena is all-reduced once per step -- the single-process loss on the concatenated batch.
"""
from __future__ import annotations

import contextlib
import itertools
import os

import torch
import torch.distributed as dist

from .. import losses
from ..distributed import gather_negatives
from ..optim import Adam
from .model import BetaVAE_H

__all__ = ["Solver"]

_MILESTONE = 50000          # iterations between numbered checkpoints (solver.py:86-87)


class _SplitPairs(torch.autograd.Function):
    """``mu -> (mu[::2], mu[1::2])`` (solver.py:64-65) as the same strided views; the backward interleaves the two gradients with ONE
    launch.  Autograd's own slice backward costs two zero fills, two strided copies and an add per step (five launches of ~5 us in a
    1.05 ms iteration, `tools/c5_step_sequence.sh`)."""

    @staticmethod
    def forward(ctx, mu):
        ctx.rows = mu.shape[0]
        return mu[::2], mu[1::2]

    @staticmethod
    def backward(ctx, g_first, g_second):
        if g_first is None or g_second is None:
            ref = g_first if g_first is not None else g_second
            g_first = torch.zeros_like(ref) if g_first is None else g_first
            g_second = torch.zeros_like(ref) if g_second is None else g_second
        return torch.stack((g_first, g_second), dim=1).flatten(0, 1)


class _WindowMean:
    """Mean of the last `every` values, emitted each time the window fills (the reference's running_loss bookkeeping)."""

    def __init__(self, every: int):
        self.every, self.total, self.count = int(every), 0.0, 0

    def push(self, value: float):
        self.total += value
        self.count += 1
        if self.count < self.every:
            return None
        mean, self.total, self.count = self.total / self.every, 0.0, 0
        return mean


class Solver(object):
    # args fields copied onto the instance under (old name -> new name)
    _FIELDS = {"ckpt_dir": "ckpt_dir", "output_dir": "output_dir", "dataset": "dataset", "max_iter": "max_iter", "z_dim": "z_dim",
               "num_channel": "nc", "lr": "lr", "beta1": "beta1", "beta2": "beta2", "ckpt_name": "ckpt_name",
               "log_step": "log_step", "save_step": "save_step"}

    def __init__(self, args, data_loader=None):
        if not (getattr(args, "cuda", False) and torch.cuda.is_available()):
            raise RuntimeError("cl_ica_amd.kitti_masks.Solver runs on the GPU only (args.cuda and a visible MI355X): "
                               "the loss / head / optimizer have no CPU path")
        for src, dst in self._FIELDS.items():
            setattr(self, dst, getattr(args, src))
        self.data_loader = data_loader
        self.global_iter = 0
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0

        self.net = BetaVAE_H(self.z_dim, self.nc, args.box_norm).to(self.device)
        if self.world > 1:          # replicas start identical whatever each rank's RNG state was
            for prm in self.net.parameters():
                dist.broadcast(prm.data, src=0)
        self.optim = Adam(self.net.parameters(), lr=self.lr, betas=(self.beta1, self.beta2))
        self.loss = losses.LpSimCLRLoss(p=args.p, tau=1.0, simclr_compatibility_mode=True)

    # ------------------------------------------------------------------ one batch
    def train_iteration(self, x):
        """One batch of image pairs (rows 2i, 2i+1): returns the 0-dim loss tensor, no host sync (solver.py:61-74)."""
        mu = self.net(x.to(self.device))
        first, second = _SplitPairs.apply(mu)                   # mu[::2], mu[1::2]: strided views, consumed as such by the loss kernels
        # negatives: the other first views.  One rank: the reference's roll; several: every rank's (order is invisible to the LSE)
        negatives = gather_negatives(first.contiguous()) if self.world > 1 else losses.RolledRows(first, 1)      # torch.roll(first, 1, 0), not materialised
        total, _, _ = self.loss(None, None, None, first, second, negatives)
        self.optim.zero_grad()
        total.backward()
        self.optim.all_reduce_grads()
        self.optim.step()
        return total

    def capture(self, x, warmup: int = 3):
        """The whole iteration on a static batch `x` as ONE HIP graph: returns ``(replay, loss)`` -- ``replay()`` runs
        forward, loss, backward and the optimizer step again on whatever ``x`` holds by then (copy the next batch into it) and refreshes
        the 0-dim tensor ``loss``.  Everything the step launches is capture-safe by construction (no host sync, device-side step counter
        in the flat Adam, the conv stack's buffers pooled); the warm-up iterations run on the capture stream first, so that the
        per-stream workspaces and the pooled buffers exist before the capture begins.  The parameters advance during the warm-up
        (`warmup` optimizer steps on `x`, exactly as if ``train_iteration`` had been called); the capture itself executes nothing, ``loss``
        holds a value after the first ``replay()``."""
        if self.world > 1:
            raise RuntimeError("Solver.capture: the data-parallel step (gloo / RCCL collectives issued from Python) is not captured")
        x = x.to(self.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.train_iteration(x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            loss = self.train_iteration(x)
        return graph.replay, loss

    # ------------------------------------------------------------------ the loop
    def _batches(self):
        """Endless stream of image batches: the loader is re-iterated epoch after epoch, labels dropped."""
        for epoch in itertools.count():
            seen = False
            for images, _ in self.data_loader:
                seen = True
                yield images
            if not seen:
                raise RuntimeError("Solver.train: the data loader yields no batches (epoch %d)" % epoch)

    def train(self):
        """Runs until ``max_iter`` iterations have been done; returns False (the reference's `failure` flag, never set there)."""
        self.net_mode(train=True)
        window = _WindowMean(self.log_step)
        # data parallel: rank 0 alone owns log.csv and the checkpoints (replicas are identical after every step); the logged
        # value is the mean over the ranks' row blocks = the loss of the global batch
        writer = self.rank == 0
        with (open(os.path.join(self.output_dir, "log.csv"), "a", 1) if writer else contextlib.nullcontext()) as log:
            if writer:
                log.write("Total Loss\n")
            for images in self._batches():
                total = self.train_iteration(images).detach()
                if self.world > 1:
                    total = total.clone()
                    dist.all_reduce(total, op=dist.ReduceOp.SUM)
                    total /= self.world
                mean = window.push(total.item())
                self.global_iter += 1
                if writer:
                    if mean is not None:
                        log.write("%.6f\n" % mean)
                    if self.global_iter % self.save_step == 0:
                        self.save_checkpoint("last")
                    if self.global_iter % _MILESTONE == 0:
                        self.save_checkpoint(str(self.global_iter))
                if self.global_iter >= self.max_iter:
                    break
        return False

    # ------------------------------------------------------------------ checkpoints (solver.py:98-128 layout)
    def _checkpoint_path(self, filename):
        return os.path.join(self.ckpt_dir, filename)

    def save_checkpoint(self, filename, silent=True):
        path = self._checkpoint_path(filename)
        payload = dict(iter=self.global_iter, model_states=dict(net=self.net.state_dict()),
                       optim_states=dict(optim=self.optim.state_dict()))
        with open(path, mode="wb+") as fh:
            torch.save(payload, fh)
        if not silent:
            print(f"=> saved checkpoint '{path}' (iter {self.global_iter})")

    def load_checkpoint(self, filename):
        path = self._checkpoint_path(filename)
        if not os.path.isfile(path):
            print(f"=> no checkpoint found at '{path}'")
            return
        payload = torch.load(path)
        self.net.load_state_dict(payload["model_states"]["net"])
        self.optim.load_state_dict(payload["optim_states"]["optim"])
        self.global_iter = payload["iter"]
        print(f"=> loaded checkpoint '{path} (iter {self.global_iter})'")

    def net_mode(self, train):
        if not isinstance(train, bool):
            raise ValueError("Only bool type is supported. True or False")
        self.net.train(train)
