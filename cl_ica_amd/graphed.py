"""HIP-graph replay of the reference's UNCHANGED ``train_step`` closure (main_mlp.py:258-285) on the drop-in modules.

The reference's loop is host-paced on this hardware: its step is ~0.35 ms of GPU work behind ~0.6-1 ms of Python, autograd bookkeeping and
kernel launches (bench.py ``dropin``: 1 340 / 1 690 steps/s against 2 700 for the fused engine).  ``capture_train_step`` records ONE call of
the closure -- forward through the drop-in encoder, loss, ``backward()``, ``optimizer.step()`` -- into a HIP graph and returns a callable
with the closure's own signature that copies the new batch into the graph's input buffers, replays it (one launch) and returns what the
closure returns.

    train_step = cl_ica_amd.capture_train_step(train_step, (z1, z2), loss, optimizer)     # once, before the loop
    ...
    total_loss_value, losses_value = train_step((z1, z2), loss, optimizer)                # every step: unchanged call site

What makes the unchanged closure capturable:
  * ``tensor.item()`` -- the closure's three host reads (main_mlp.py:283-285, ``unpack_item_list``) -- cannot run inside a capture.  While
    the closure is being recorded ``Tensor.item`` answers with a placeholder and remembers the tensor; after every replay the placeholders
    in the closure's return value are replaced by that replay's floats.
  * the scalars a drop-in LOSS returns are final right behind the loss forward, a third of the way into the step.  The loss call records a
    one-wave kernel there (``clica_publish_host``) that writes them to pinned host memory followed by a sequence number; the replaying
    callable spins on the sequence number and returns while the backward pass and the optimizer of the same replay are still running, so
    the host samples and queues the next batch behind them and the device never idles (bench.py ``dropin.captured``).  Scalars of any
    other origin are gathered at the end of the graph and cost one stream synchronisation per step.
  * everything else the step does on the device already goes through stream-ordered launches on preallocated or graph-pool memory: the
    deferred stacking of the two encoder calls, the roll detection of the loss, the optimizer hooks that re-pack the weights at step end.
  * the optimizer must be capturable: ``cl_ica_amd.optim.Adam`` (device-side step counter) or ``torch.optim.Adam(..., capturable=True)``.
Limits: tensor SHAPES are fixed at capture (a ragged last batch: call the original closure); host-side control flow inside the closure is
frozen as recorded; ``.item()`` is the only host read the recorder understands (``float(t)``, ``t.cpu()``, ``print(t)`` inside the closure
raise at capture, as any synchronisation does).
"""
from __future__ import annotations

import contextlib
import functools
import time
from typing import Any, Callable, List, Optional

import torch

__all__ = ["capture_train_step"]

_RECORDER: Optional["_Recorder"] = None


def active_recorder() -> Optional["_Recorder"]:
    """The recorder of the `capture_train_step` call in progress (losses._share_items asks), else None."""
    return _RECORDER


class _EarlySlot:
    """Placeholder for a scalar published from the middle of the graph (publisher index, position)."""
    __slots__ = ("pub", "k")

    def __init__(self, pub: int, k: int):
        self.pub, self.k = pub, k


class _Publisher:
    """Buffers of one mid-graph host read, allocated BEFORE the capture (allocations are not capturable)."""
    __slots__ = ("host", "host_seq", "seq_dev", "seq_np", "n")

    def __init__(self, device):
        self.n = 0
        self.host = torch.zeros(64, dtype=torch.float32, pin_memory=True)
        self.host_seq = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        self.seq_np = self.host_seq.numpy()
        self.seq_dev = torch.zeros(1, dtype=torch.int32, device=device)

    def wait(self, expected: int, device) -> List[float]:
        seq, t0, spins = self.seq_np, None, 0
        want = expected & 0x7FFFFFFF
        while (int(seq[0]) & 0x7FFFFFFF) != want:
            spins += 1
            if spins % 4096 == 0:
                now = time.monotonic()
                t0 = t0 or now
                if now - t0 > 20.0:
                    torch.cuda.current_stream(device).synchronize()      # surfaces a device fault as the error it is
                    if (int(seq[0]) & 0x7FFFFFFF) != want:
                        raise RuntimeError(f"captured train_step: the loss scalars of replay {expected} never arrived (sequence {int(seq[0])})")
        return self.host[:self.n].tolist()


class _Recorder:
    MAX_PUBLISHERS = 4       # loss calls per closure that get the early read; further ones are gathered at the end of the graph

    def __init__(self, device):
        self.spare = [_Publisher(device) for _ in range(self.MAX_PUBLISHERS)]
        self.publishers: List[_Publisher] = []

    def publish(self, src: torch.Tensor, outs) -> None:
        """Called by the loss forward under capture: src = the adjacent fp32 scalars behind `outs`."""
        src = src.detach().reshape(-1)
        if src.dtype != torch.float32 or not src.is_contiguous() or src.numel() > 64 or not self.spare:
            return
        from . import ops
        p = self.spare.pop()
        p.n = src.numel()
        ops.publish_host(src, p.host, p.seq_dev, p.host_seq)
        self.publishers.append(p)
        pub = len(self.publishers) - 1
        for k, t in enumerate(outs):
            t.item = functools.partial(_EarlySlot, pub, k)


class _Slot:
    """Placeholder for a host scalar read inside the recorded closure."""
    __slots__ = ("index",)

    def __init__(self, index: int):
        self.index = index

    def __repr__(self):
        return f"<deferred .item() #{self.index}>"


def _tree_map(fn: Callable[[Any], Any], x: Any) -> Any:
    if isinstance(x, (list, tuple)):
        mapped = [_tree_map(fn, v) for v in x]
        return type(x)(mapped) if not hasattr(x, "_fields") else type(x)(*mapped)
    if isinstance(x, dict):
        return {k: _tree_map(fn, v) for k, v in x.items()}
    return fn(x)


@contextlib.contextmanager
def _recording_item(slots: List[torch.Tensor]):
    orig = torch.Tensor.item

    def item(self):
        if self.is_cuda and torch.cuda.is_current_stream_capturing():
            if self.numel() != 1:
                raise ValueError("only one-element tensors can be converted to Python scalars")
            slots.append(self.detach().reshape(()))
            return _Slot(len(slots) - 1)
        return orig(self)
    torch.Tensor.item = item
    try:
        yield
    finally:
        torch.Tensor.item = orig


def capture_train_step(train_step: Callable, data, *args, warmup: int = 3, **kwargs) -> Callable:
    """Record ``train_step(data, *args, **kwargs)`` into a HIP graph (after `warmup` eager calls on a side stream, which DO train: they are
    ordinary steps on `data`) and return ``replay(data, *args, **kwargs) -> the closure's return value``.  ``data``: any nesting of
    tuples / lists / dicts of CUDA tensors (the reference passes ``(z1, z2_con_z1)``); the other arguments are bound as given."""
    tensors = []
    _tree_map(lambda t: tensors.append(t) if isinstance(t, torch.Tensor) else None, data)
    if not tensors or not all(t.is_cuda for t in tensors):
        raise ValueError("capture_train_step: `data` must hold CUDA tensors")
    device = tensors[0].device
    static = _tree_map(lambda t: t.detach().clone() if isinstance(t, torch.Tensor) else t, data)
    static_leaves: List[torch.Tensor] = []
    _tree_map(lambda t: static_leaves.append(t) if isinstance(t, torch.Tensor) else None, static)

    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        for _ in range(max(1, warmup)):
            train_step(static, *args, **kwargs)
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)

    global _RECORDER
    if _RECORDER is not None:
        raise RuntimeError("capture_train_step: another capture is in progress")
    graph = torch.cuda.CUDAGraph()
    slots: List[torch.Tensor] = []
    rec = _RECORDER = _Recorder(device)
    try:
        # (the recorded call sees the batch as RollDeferring views of the static buffers: a torch.roll(z1, 1, 0) nobody reads -- the
        #  reference's z3, main_mlp.py:266 -- then records no launch; anything that does read it gets the rolled tensor)
        from .lazy import RollDeferring
        static_view = _tree_map(lambda t: t.as_subclass(RollDeferring) if isinstance(t, torch.Tensor) and not t.requires_grad else t, static)
        with _recording_item(slots):
            with torch.cuda.graph(graph, stream=side):
                recorded = train_step(static_view, *args, **kwargs)
                if slots:                                        # the gather of the host scalars is part of the graph: one D2H copy per replay
                    stacked_dtype = torch.float64 if any(s.dtype == torch.float64 for s in slots) else torch.float32
                    dev_buf = torch.stack([s.to(stacked_dtype) for s in slots])
    finally:
        _RECORDER = None
    if slots:
        host = torch.empty(len(slots), dtype=stacked_dtype, pin_memory=True)
    replays = [0]
    shapes = [tuple(t.shape) for t in static_leaves]

    def replay(new_data, *a, **k):
        leaves: List[torch.Tensor] = []
        _tree_map(lambda t: leaves.append(t) if isinstance(t, torch.Tensor) else None, new_data)
        if len(leaves) != len(static_leaves) or any(tuple(t.shape) != s for t, s in zip(leaves, shapes)):
            return train_step(new_data, *a, **k)             # another batch shape than the recorded one: the closure itself
        for dst, src in zip(static_leaves, leaves):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        graph.replay()
        replays[0] += 1
        early = [p.wait(replays[0], device) for p in rec.publishers]      # returns behind the loss forward of THIS replay
        vals = None
        if slots:
            host.copy_(dev_buf, non_blocking=True)
            torch.cuda.current_stream(device).synchronize()
            vals = host.tolist()

        def fill(v):
            if isinstance(v, _EarlySlot):
                return early[v.pub][v.k]
            if isinstance(v, _Slot):
                return vals[v.index]
            return v
        return _tree_map(fill, recorded)

    replay.graph = graph
    replay.static_inputs = static
    replay.n_host_scalars = len(slots) + sum(p.n for p in rec.publishers)
    replay.n_early_scalars = sum(p.n for p in rec.publishers)
    return replay
