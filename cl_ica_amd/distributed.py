"""Data-parallel pieces (one process per GPU; backend "nccl" == RCCL over xGMI on ROCm, "gloo" in
the CPU tests).  The reference has no distributed code; the semantics to preserve are those of its
single-process loss on the gathered batch (nn.DataParallel precedent, main_3dident.py:373).

* ``gather_negatives``: autograd-aware all-gather of the local embeddings -> global negatives pool;
  backward = reduce-scatter(sum) of the pool's gradient (a plain all_gather would drop the
  cross-rank gradient and not match the single-process reference).
* ``GradBuckets``: bucketed asynchronous all-reduce(sum) of the flat gradient arena, launched as
  backward finishes each layer so RCCL overlaps the remaining wgrad/dgrad GEMMs; the 1/world
  average is folded into the fused Adam (``grad_scale``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

__all__ = ["gather_negatives", "GradBuckets", "init_from_env"]


class _GatherNegatives(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, group):
        ctx.group = group
        world = dist.get_world_size(group)
        out = torch.empty((world * z.shape[0], z.shape[1]), dtype=z.dtype, device=z.device)
        dist.all_gather_into_tensor(out, z.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        world = dist.get_world_size(ctx.group)
        out = torch.empty((g.shape[0] // world, g.shape[1]), dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(out, g.contiguous(), op=dist.ReduceOp.SUM, group=ctx.group)
        return out, None


def gather_negatives(z: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """All ranks' rows of ``z`` concatenated in rank order, differentiable w.r.t. the local rows."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return z
    return _GatherNegatives.apply(z, group)


class GradBuckets:
    """Groups consecutive per-layer slices of a flat gradient arena (given in backward completion
    order) into buckets of ~``bucket_bytes`` and all-reduces each bucket as soon as its last layer's
    gradient has been written."""

    def __init__(self, arena: torch.Tensor, layer_slices: Sequence[Tuple[int, int]], world: int,
                 group: Optional[dist.ProcessGroup], bucket_bytes: int = 8 << 20, force: bool = False,
                 boundaries: Sequence[int] = ()):
        self.arena, self.group, self.world = arena, group, world
        self.force = force
        self.buckets: List[Tuple[int, int]] = []      # (lo, hi) element ranges
        self.trigger: dict = {}                        # layer index (completion order) -> bucket id
        lo = hi = None
        for i, (a, b) in enumerate(layer_slices):
            lo = a if lo is None else min(lo, a)
            hi = b if hi is None else max(hi, b)
            # `boundaries`: completion indices after which a bucket closes whatever its size (the engine's two-half weight-gradient
            # launch: the first half's all-reduce then runs under the second half's GEMMs)
            if (hi - lo) * 4 >= bucket_bytes or i == len(layer_slices) - 1 or i in boundaries:
                self.trigger[i] = len(self.buckets)
                self.buckets.append((lo, hi))
                lo = hi = None
        self.pending: List = []
        self.comm_stream = torch.cuda.Stream(device=arena.device) if arena.is_cuda else None

    def layer_done(self, i: int):
        b = self.trigger.get(i)
        if b is None or (self.world == 1 and not self.force):
            return
        lo, hi = self.buckets[b]
        view = self.arena[lo:hi]
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream(view.device))
            with torch.cuda.stream(self.comm_stream):
                self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending.clear()
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.arena.device).wait_stream(self.comm_stream)


def init_from_env(backend: Optional[str] = None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world, device)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tools/dp_bench_smoke.sh, tests/test_gpu_bench.py: the multi-rank control flow of bench.py on a ONE-GPU box):
    # CLICA_DIST_BACKEND=gloo runs the collectives over gloo, and ranks beyond the box's GPUs share cuda:0 (RCCL refuses two ranks on one device)
    backend = backend or os.environ.get("CLICA_DIST_BACKEND") or None
    if backend == "gloo" and torch.cuda.is_available() and local >= torch.cuda.device_count():
        local = 0
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if device.type == "cuda" else "gloo"), rank=rank, world_size=world,
                                device_id=device if (device.type == "cuda" and (backend or "nccl") == "nccl") else None)
    return rank, world, device
