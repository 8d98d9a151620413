"""Deferred encoder outputs: let two calls of the same module share ONE launch.

The reference's train_step calls the encoder twice per step on batches of B rows (main_mlp.py:270-271:
``z1_rec = h(z1); z2_con_z1_rec = h(z2_con_z1)``).  The whole-encoder kernels own 48 rows per workgroup for the whole stack, so a
B = 6144 call fills 128 of the 256 CUs and costs as much as a 12 288-row call: through the drop-in modules the reference's loop
paid every encoder phase twice.  With this module ``FusedMLP.forward`` returns a ``LazyOut`` -- a
``torch.Tensor`` subclass that knows its shape / dtype / device but has no storage yet.  The first thing that happens to it decides:

  * the same module is called again with a batch of the same width (the reference's second ``h(.)``): both inputs are stacked,
    the stack runs as ONE autograd node (one forward launch, later one backward-chain and one weight-gradient launch), and the
    two results are row slices of its output;
  * anything else touches it (any torch function, an attribute beyond the metadata, our own losses / layers): the pending call is
    computed on its own, exactly as before.

Either way the values are the module's values for those inputs (rows of an MLP are independent) and the autograd graph is an
ordinary one: a ``LazyOut`` never enters a graph, every consumer receives the materialised plain tensor (``__torch_function__``).
The parameters must not be modified in place between the call and the first use (checked, as autograd checks saved tensors).
``CLICA_DROPIN_LAZY=0`` turns the mechanism off.

This file has no device code; the mechanics are covered on CPU (tests/test_host_logic.py) with a stand-in compute function.
"""
from __future__ import annotations

import contextlib
import os
import weakref
from typing import Callable, List, Optional

import torch
from torch.utils._pytree import tree_map

__all__ = ["LazyOut", "LazyRoll", "RollDeferring", "defer", "plain", "enabled", "flush_all", "after_step", "AFTER_STEP", "attach", "find_optimizers"]

_META_GETTERS = {"shape", "dtype", "device", "requires_grad", "ndim", "layout", "is_cuda", "is_leaf_placeholder"}
_META_METHODS = {"dim", "size", "__len__", "ndimension", "numel", "nelement", "is_floating_point", "is_complex", "get_device",
                 "element_size", "is_contiguous_placeholder"}
_ALL_PENDING = weakref.WeakSet()


def enabled() -> bool:
    return os.environ.get("CLICA_DROPIN_LAZY", "1") != "0"


class _Pending:
    """Calls of one owner (module) that have not run yet: at most `max_items` inputs of the same trailing shape."""

    def __init__(self, owner, compute: Callable[[torch.Tensor], torch.Tensor], params, max_items: int = 2, compute_many=None):
        self.owner, self.compute, self.max_items = owner, compute, max_items
        self.compute_many = compute_many       # optional: list of inputs -> list of outputs (or None) as one autograd node
        self.params = list(params)
        self.versions = [p._version for p in self.params]
        self.items: List = []          # (input tensor, LazyOut)
        self.in_versions: List[int] = []      # _version of every input at the moment of its call (an input mutated in place
                                              # between the call and the first use would silently give the output for the NEW values)
        self.grad_mode = torch.is_grad_enabled()
        self.stream = torch.cuda.current_stream() if torch.cuda.is_available() else None      # the deferred launches belong to this stream
        _ALL_PENDING.add(self)

    def add(self, x: torch.Tensor, lz) -> None:
        self.items.append((x, lz))
        self.in_versions.append(x._version)

    def stale(self) -> bool:
        if any(p._version != v for p, v in zip(self.params, self.versions)):
            return True
        return any(x._version != v for (x, _), v in zip(self.items, self.in_versions))

    def compatible(self, x: torch.Tensor) -> bool:
        if not self.items or len(self.items) >= self.max_items or self.stale():
            return False
        x0 = self.items[0][0]
        return x.shape[1:] == x0.shape[1:] and x.dtype == x0.dtype and x.device == x0.device

    def flush(self):
        """Run the pending calls (one compute over the stacked inputs) and hand every LazyOut its rows.  Returns the list of results in
        call order (None if the calls had gone stale)."""
        was_stale = self.stale()
        items, self.items = self.items, []
        self.in_versions = []
        _ALL_PENDING.discard(self)
        if getattr(self.owner, "_clica_pending", None) is self:
            self.owner._clica_pending = None
        if not items:
            return None
        if was_stale:
            # a parameter was written in place (not by an optimizer: those flush first, see the step pre-hook below) while the call
            # was pending: its value for the OLD parameters can no longer be computed.  Nobody may have wanted it (a discarded
            # evaluation call); whoever does gets the error.
            for _, lz in items:
                if lz is not None:
                    lz._stale = True
            _rearm_search(self.owner)      # (an optimizer this module has not found yet may have written them: look again at the next call)
            return None
        # the call's own grad mode and stream, not those of the first use (which may sit inside a no_grad block or on another stream)
        cur = torch.cuda.current_stream() if self.stream is not None else None
        other = self.stream is not None and cur != self.stream
        if other:
            self.stream.wait_stream(cur)       # inputs the current stream produced since
        with torch.set_grad_enabled(self.grad_mode), (torch.cuda.stream(self.stream) if other else contextlib.nullcontext()):
            if len(items) == 1:
                vals = [self.compute(items[0][0])]
            else:
                vals = self.compute_many([x for x, _ in items]) if self.compute_many is not None else None
                if vals is None:
                    ystack = self.compute(torch.cat([x for x, _ in items], 0))
                    vals, off = [], 0
                    for x, _ in items:
                        vals.append(ystack[off:off + x.shape[0]])
                        off += x.shape[0]
        if other:
            cur.wait_stream(self.stream)
        for (_, lz), v in zip(items, vals):
            if lz is not None:
                lz._value = v
        return vals


class LazyOut(torch.Tensor):
    """Placeholder for the output of a deferred module call (see the module docstring)."""

    @staticmethod
    def __new__(cls, pending: _Pending, shape, dtype, device, requires_grad: bool):
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=dtype, device=device, requires_grad=requires_grad)
        r._pending = pending
        r._value = None
        r._stale = False
        return r

    def materialize(self) -> torch.Tensor:
        if self._value is None and not self._stale:
            self._pending.flush()
        if self._stale:
            raise RuntimeError("cl_ica_amd: a parameter of the encoder or the input of the call was modified in place between the call and the "
                               "first use of its (deferred) output; use the output first, or set CLICA_DROPIN_LAZY=0")
        return self._value

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name == "__get__" and getattr(getattr(func, "__self__", None), "__name__", "") in _META_GETTERS or name in _META_METHODS:
            with torch._C.DisableTorchFunctionSubclass():      # metadata lives on the wrapper itself: no need to compute anything
                return func(*args, **kwargs)

        if name == "roll" and args and type(args[0]) is LazyOut:
            # z3_rec = torch.roll(z1_rec, 1, 0) (main_mlp.py:272): LpSimCLRLoss never reads the rolled copy (losses.py: _PairLossSymFn), so the
            # roll itself is deferred as well -- any other consumer gets the real tensor through LazyRoll.materialize()
            shift = _roll_rows_args(args, kwargs)
            if shift is not None:
                return LazyRoll(args[0], shift)

        with torch._C.DisableTorchFunctionSubclass():
            return func(*tree_map(_unwrap, args), **tree_map(_unwrap, kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):      # anything that slipped past __torch_function__
        kwargs = kwargs or {}
        return func(*tree_map(_unwrap, args), **tree_map(_unwrap, kwargs))


class LazyRoll(torch.Tensor):
    """``torch.roll(source, shift, 0)`` of a deferred module output, itself deferred: the reference's ``z3_rec`` (main_mlp.py:272).
    ``LpSimCLRLoss`` recognises it by ``source`` and never computes it; every other consumer (any torch function, ``plain``) gets the
    real rolled tensor, computed once with ordinary autograd."""

    @staticmethod
    def __new__(cls, source, shift: int):
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(source.shape), dtype=source.dtype, device=source.device,
                                                requires_grad=source.requires_grad)
        r.source, r.shift, r._value = source, int(shift), None
        return r

    def materialize(self) -> torch.Tensor:
        if self._value is None:
            self._value = torch.roll(plain(self.source), self.shift, 0)
        return self._value

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name == "__get__" and getattr(getattr(func, "__self__", None), "__name__", "") in _META_GETTERS or name in _META_METHODS:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*tree_map(_unwrap, args), **tree_map(_unwrap, kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        return func(*tree_map(_unwrap, args), **tree_map(_unwrap, kwargs))


def _unwrap(a):
    return a.materialize() if isinstance(a, (LazyOut, LazyRoll)) else a


def _roll_rows_args(args, kwargs):
    """(shift) of a ``roll(x, s, 0)`` call with one integer shift along dim 0, else None."""
    rest = list(args[1:])
    shifts = kwargs.get("shifts", rest[0] if rest else None)
    dims = kwargs.get("dims", rest[1] if len(rest) > 1 else None)
    dims = dims[0] if isinstance(dims, (tuple, list)) and len(dims) == 1 else dims
    shifts = shifts[0] if isinstance(shifts, (tuple, list)) and len(shifts) == 1 else shifts
    return shifts if isinstance(shifts, int) and not isinstance(shifts, bool) and dims == 0 and len(args[0].shape) >= 1 else None


class RollDeferring(torch.Tensor):
    """An ordinary tensor (same storage: ``t.as_subclass(RollDeferring)``) on which ``torch.roll(t, s, 0)`` is deferred (a LazyRoll): what
    ``capture_train_step`` hands the closure as its batch, so that ``z3 = torch.roll(z1, 1, 0)`` (main_mlp.py:266) -- which the losses of
    this package never read -- records no launch.  Every other operation runs on the plain tensor and returns plain tensors."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if getattr(func, "__name__", "") == "roll" and args and type(args[0]) is RollDeferring:
            shift = _roll_rows_args(args, kwargs)
            if shift is not None:
                return LazyRoll(args[0].as_subclass(torch.Tensor), shift)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def plain(t):
    """The plain tensor behind `t` (computing it if it is still pending); anything that is not a LazyOut is returned as is.
    Every entry point of this package that takes embeddings calls this first: a LazyOut must never reach autograd.Function.apply."""
    return t.materialize() if isinstance(t, (LazyOut, LazyRoll)) else t


def defer(owner, x: torch.Tensor, compute: Callable[[torch.Tensor], torch.Tensor], out_shape, params, max_items: int = 2,
          compute_many=None):
    """Register the call `compute(x)` of `owner`.  If `owner` has a pending call with a compatible input the two are stacked and
    run NOW (one launch); the result of this call is returned as a plain tensor.  Otherwise a LazyOut is returned."""
    _attach_owners(owner, params)
    pend: Optional[_Pending] = getattr(owner, "_clica_pending", None)
    if pend is not None and pend.compatible(x):
        pend.add(x, None)              # the call that completes the stack needs no placeholder: its rows are returned right away
        return pend.flush()[-1]
    if pend is not None:
        pend.flush()                   # an incompatible (or stale) first call: it runs (or is dropped) on its own
    pend = _Pending(owner, compute, params, max_items, compute_many)
    owner._clica_pending = pend
    lz = LazyOut(pend, out_shape, x.dtype, x.device, True)
    pend.add(x, lz)
    return lz


HOOK_EPOCH = 0      # number of optimizer steps that went through this module's pre-hook (see _attach_owners)


def flush_all(*_args, **_kwargs):
    """Run every pending call.  Registered as an optimizer-step pre-hook on the attached optimizers (and called by cl_ica_amd.optim.Adam):
    a deferred output is always computed with the parameters of the moment the call was made."""
    global HOOK_EPOCH
    HOOK_EPOCH += 1
    for p in list(_ALL_PENDING):
        p.flush()


AFTER_STEP: List[Callable] = []      # what modules of this package want done once the parameters have changed (weight re-packing)


def after_step(*_args, **_kwargs):
    """End of an optimizer step (global post-hook below; cl_ica_amd.optim.Adam calls it itself): work that depends only on the new
    parameters runs here, where the host is about to wait for the device, instead of in front of the next step's first launch."""
    for f in AFTER_STEP:
        f()


# ---- optimizer hooks, per INSTANCE (round 5) --------------------------------------------------------------------------------------
# Rounds 3-4 registered flush_all / after_step as GLOBAL optimizer step hooks at import: every optimizer of the process ran them.  Now
# only the optimizers that actually hold parameters of a deferring module get them, as instance-level hooks
# (Optimizer.register_step_pre_hook / register_step_post_hook).  The module does not see the optimizer being built
# (`torch.optim.Adam(f.parameters(), lr)`, main_mlp.py:312, consumes a generator), so the owners are looked up once, from the module's
# first deferred calls, through the garbage collector's referrer graph: parameter <- param_group["params"] list <- param group dict <-
# optimizer.param_groups list <- optimizer.  cl_ica_amd.optim.Adam needs none of this (its step() calls both functions itself).
# Without an attached optimizer nothing breaks: a deferred output still pending when the parameters change raises on use (the staleness
# check above) instead of having been flushed, and the weight copies are re-packed in front of the next forward instead of at step end.
_ATTACHED = weakref.WeakSet()


def attach(optimizer) -> bool:
    """Give `optimizer` the two step hooks of this module (idempotent).  Returns True if it was newly attached."""
    if optimizer in _ATTACHED or not hasattr(optimizer, "register_step_pre_hook"):
        return False
    optimizer.register_step_pre_hook(flush_all)
    optimizer.register_step_post_hook(after_step)
    _ATTACHED.add(optimizer)
    return True


def find_optimizers(params) -> list:
    """The torch.optim.Optimizer instances whose param_groups hold one of `params` (see above); a few gc.get_referrers walks, started from
    the parameter's direct referrers only (lists that contain it: an optimizer's param_group["params"] is one)."""
    import gc
    if not params:
        return []
    found = []
    for p0 in params[:1] + params[-1:]:          # first and last parameter: an optimizer over a sub-set of the module is found too
        _find_from(p0, found, gc)
    return found


def _find_from(p0, found, gc) -> None:
    for lst in gc.get_referrers(p0):
        if not isinstance(lst, list) or not any(q is p0 for q in lst):
            continue
        for grp in gc.get_referrers(lst):
            if not (isinstance(grp, dict) and grp.get("params") is lst):
                continue
            for groups in gc.get_referrers(grp):
                if not (isinstance(groups, list) and any(g is grp for g in groups)):
                    continue
                for od in gc.get_referrers(groups):
                    if not (isinstance(od, dict) and od.get("param_groups") is groups):
                        continue
                    for o in gc.get_referrers(od):
                        if isinstance(o, torch.optim.Optimizer) and getattr(o, "__dict__", None) is od and not any(o is f for f in found):
                            found.append(o)


_SEARCH_TRIES = 3


def _rearm_search(owner) -> None:
    st = getattr(owner, "_clica_opt_search", None)
    if st is not None:
        st[0], st[1] = 0, False


def _attach_owners(owner, params):
    """Look the owners of `params` up (at most _SEARCH_TRIES deferred calls in a row) and attach the step hooks.  The search is RE-ARMED
     whenever the evidence says an optimizer without the hooks is at work: a pending output went stale (flush), or the
    parameters' versions moved since this module's last deferred call although no hooked optimizer step happened in between, or every
    optimizer found so far has been garbage-collected -- an optimizer built later (re-created after a learning-rate change, built after a
    few evaluation calls) is then found at the next call."""
    st = getattr(owner, "_clica_opt_search", None)
    params = list(params)
    if st is None:
        # tries, found, parameter versions at the last call, HOOK_EPOCH at the last call, the optimizers found for this owner (weak)
        st = [0, False, None, HOOK_EPOCH, weakref.WeakSet()]
        try:
            object.__setattr__(owner, "_clica_opt_search", st)
        except Exception:
            return
    vers = sum(int(q._version) for q in params)
    if st[2] is not None and vers != st[2] and HOOK_EPOCH == st[3]:
        st[0], st[1] = 0, False                       # somebody stepped without our hooks
    if st[1] and len(st[4]) == 0:
        st[0], st[1] = 0, False                       # every optimizer found for this module is gone (re-created: look for its successor)
    st[2], st[3] = vers, HOOK_EPOCH
    if st[1] or st[0] >= _SEARCH_TRIES:
        return
    st[0] += 1
    for o in find_optimizers(params):
        attach(o)
        st[4].add(o)
        st[1] = True
