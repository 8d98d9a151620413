"""Contrastive losses with the reference's call surface, computed by the HIP kernels.

Drop-in for /root/reference/losses.py: ``LpSimCLRLoss`` (losses.py:405-477) and ``SimCLRLoss``
(losses.py:162-202) keep constructor arguments, defaults, the 6-argument ``__call__`` protocol of
``CLLoss`` (losses.py:28-29; the first three arguments are ignored and may be ``None``) and the
return triple ``(mean, per_item, [pos_mean, neg_mean])``, every element autograd-connected.

Unlike the reference nothing of size B x B3 is ever materialised: forward and backward are tiled
all-pairs kernels with an online log-sum-exp (cl_ica_amd/csrc/lp_kernels.h).
"""
from __future__ import annotations

import collections
import ctypes as C
import functools
from abc import ABC, abstractmethod
from typing import Optional

import torch

from . import _lib, lazy

__all__ = ["CLLoss", "ConditionalPairCLLoss", "MarginalPairCLLoss", "LpSimCLRLoss", "SimCLRLoss", "UniformityLoss", "AlignmentLoss",
           "AlignmentUniformityLoss"]


class CLLoss(ABC):
    """Loss protocol of the reference (losses.py:11-29): one positive and one negative pair."""

    @abstractmethod
    def loss(self, z1, z2_con_z1, z3, z1_rec, z2_con_z1_rec, z3_rec):
        ...

    def __call__(self, z1, z2_con_z1, z3, z1_rec, z2_con_z1_rec, z3_rec):
        return self.loss(z1, z2_con_z1, z3, z1_rec, z2_con_z1_rec, z3_rec)


PATHS = collections.Counter()     # which backward ran (tests assert the one-sweep path is the one the reference's train_step gets)


class _SharedScalars:
    """The scalars a loss call returns, fetched ONCE and without draining the stream.

    The reference's train_step ends with ``total_loss_value.item()`` and ``[v.item() for v in losses_value]`` (main_mlp.py:283-285):
    three blocking 4-byte copies, each a full synchronisation of the stream -- behind the backward pass and the optimizer step that
    were queued in the meantime, with the device idle while the host then prepares the next step (tools/dropin_timeline.py: ~200 us
    of a 0.9 ms step).  The three values are adjacent floats of ONE result buffer that nothing writes after the loss forward, so the
    loss call itself queues one asynchronous copy of them into pinned host memory right behind the forward kernels and records an
    event; ``.item()`` waits for THAT event and answers from the host copy.  The backward / optimizer kernels of this step keep running
    while the host samples the next batch.  Installed as an INSTANCE attribute ``item`` of the returned tensors, which stay ordinary
    tensors; a tensor written in place after the call falls back to ``Tensor.item``.  ``CLICA_DROPIN_ASYNC_ITEM=0``: fetch on first
    use with an ordinary blocking copy (still one instead of three)."""

    __slots__ = ("src", "vals", "host", "event")

    def __init__(self, src):
        self.src, self.vals, self.host, self.event = src, None, None, None
        if src.is_cuda and _async_item() and not torch.cuda.is_current_stream_capturing():      # (an event of a captured stream cannot be waited for)
            self.host = torch.empty(src.numel(), dtype=src.dtype, pin_memory=True)
            self.host.copy_(src, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()

    def __getstate__(self):      # copy.deepcopy / torch.save of a returned loss tensor carry this object along: keep only the numbers
        if self.vals is None:
            if self.event is not None:
                self.event.synchronize()
                self.vals = self.host.tolist()
            else:
                self.vals = self.src.tolist()
        return {"vals": self.vals}

    def __setstate__(self, state):
        self.src, self.host, self.event, self.vals = None, None, None, state["vals"]

    def item(self, t, k, version):
        if t._version != version:
            return torch.Tensor.item(t)
        if self.vals is None:
            if self.event is not None:
                self.event.synchronize()
                self.vals = self.host.tolist()
            else:
                self.vals = self.src.tolist()
        return self.vals[k]


def _async_item() -> bool:
    """The three `.item()` calls of the reference's train_step share one asynchronous copy behind the loss forward (always on)."""
    return True


def _share_items(src, outs):
    if src.is_cuda and torch.cuda.is_current_stream_capturing():
        # tensors of a captured graph are rewritten by every replay: they keep the plain Tensor.item (a cached value would go stale) --
        # unless cl_ica_amd.capture_train_step is recording, which publishes them to the host from this point of the graph
        from . import graphed
        rec = graphed.active_recorder()
        if rec is not None:
            rec.publish(src, outs)
        return
    sh = _SharedScalars(src)
    for k, t in enumerate(outs):
        t.item = functools.partial(sh.item, t, k, t._version)
        # (a reference cycle tensor -> partial -> tensor: collected by the cycle GC like any other; the tensors are 4-byte views)


def _means_of(mean):
    """The 3-float buffer (mean, pos_mean, neg_mean) the forward wrote, recovered from its first element (a 0-dim view at offset 3B)."""
    return torch.as_strided(mean.detach(), (3,), (1,))


def _prep(name, t):
    if t.dim() != 2:
        raise ValueError(f"{name} must be 2-D (batch, n), got shape {tuple(t.shape)}")
    _lib.require_cuda(t, name)
    return _lib.rowmajor(t.detach())


class _PairLossFn(torch.autograd.Function):
    """autograd bridge to clica_{lp,dot}_loss_{fwd,bwd}; `desc` is the ctypes descriptor."""

    @staticmethod
    def forward(ctx, z1, z2, z3, kind, desc):
        ctx.set_materialize_grads(False)      # an output nobody differentiates arrives as None in backward, not as a tensor of zeros
        lib = _lib.load()
        (a, lda), (b, ldb), (c, ldc) = _prep("z1_rec", z1), _prep("z2_con_z1_rec", z2), _prep("z3_rec", z3)
        B = a.shape[0]
        out = torch.empty(3 * B + 3, dtype=torch.float32, device=a.device)
        loss_i, pos_i, lse_i, means = out[:B], out[B:2 * B], out[2 * B:3 * B], out[3 * B:]
        fwd_b, bwd_b = C.c_size_t(), C.c_size_t()
        ws_query = lib.clica_lp_loss_workspace_bytes if kind == "lp" else lib.clica_dot_loss_workspace_bytes
        _lib.check(ws_query(C.byref(desc), C.byref(fwd_b), C.byref(bwd_b)), f"clica_{kind}_loss_workspace_bytes")
        ws = _lib.workspace(f"{kind}_loss", max(fwd_b.value, bwd_b.value), a.device)
        fwd = lib.clica_lp_loss_fwd if kind == "lp" else lib.clica_dot_loss_fwd
        # when z1 will need a gradient, let the forward sweep also accumulate the softmax-weighted row
        # gradient: the backward then needs only the column sweep
        n = a.shape[1]
        # (autograd.Function.forward runs with grad mode off: ask the node, not torch.is_grad_enabled())
        want_rg = ctx.needs_input_grad[0] and not (kind == "lp" and desc.p < 1.0 and not desc.no_eps)
        rowgrad = torch.empty((B, n), dtype=torch.float32, device=a.device) if want_rg else None
        _lib.check(fwd(C.byref(desc), a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), ldc,
                       loss_i.data_ptr(), pos_i.data_ptr(), lse_i.data_ptr(), means.data_ptr(),
                       _lib.ptr(rowgrad), n, ws.data_ptr(), ws.numel(), _lib.stream_ptr()), f"clica_{kind}_loss_fwd")
        ctx.rowgrad = rowgrad
        ctx.save_for_backward(a, b, c, lse_i)
        ctx.lds = (lda, ldb, ldc)
        ctx.kind, ctx.desc = kind, desc
        ctx.shapes = (z1.shape, z2.shape, z3.shape)
        mean, pos_mean, neg_mean = means.unbind(0)
        return mean, loss_i, pos_mean, neg_mean

    @staticmethod
    def backward(ctx, g_mean, g_item, g_pos, g_neg):
        lib = _lib.load()
        a, b, c, lse_i = ctx.saved_tensors
        lda, ldb, ldc = ctx.lds
        kind, desc = ctx.kind, ctx.desc
        dev = a.device
        need1, need2, need3 = ctx.needs_input_grad[:3]

        def scal(g):  # 0-dim upstream gradient -> contiguous fp32 device scalar (None stays None)
            return None if g is None else g.detach().to(torch.float32).reshape(1).contiguous()
        g_mean_t = scal(g_mean)
        if g_mean_t is None:   # the C ABI reads NULL as 1.0; an unused mean output must weigh 0
            g_mean_t = torch.zeros(1, dtype=torch.float32, device=dev)
        g_item_t = None if g_item is None else g_item.detach().to(torch.float32).contiguous()
        g_pos_t, g_neg_t = scal(g_pos), scal(g_neg)
        n = a.shape[1]
        dz1 = torch.empty((a.shape[0], n), dtype=torch.float32, device=dev) if need1 else None
        dz2 = torch.empty((b.shape[0], n), dtype=torch.float32, device=dev) if need2 else None
        dz3 = torch.empty((c.shape[0], n), dtype=torch.float32, device=dev) if need3 else None
        fwd_b, bwd_b = C.c_size_t(), C.c_size_t()
        ws_query = lib.clica_lp_loss_workspace_bytes if kind == "lp" else lib.clica_dot_loss_workspace_bytes
        _lib.check(ws_query(C.byref(desc), C.byref(fwd_b), C.byref(bwd_b)), f"clica_{kind}_loss_workspace_bytes")
        ws = _lib.workspace(f"{kind}_loss", max(fwd_b.value, bwd_b.value), dev)
        bwd = lib.clica_lp_loss_bwd if kind == "lp" else lib.clica_dot_loss_bwd
        PATHS["generic"] += 1
        _lib.check(bwd(C.byref(desc), a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), ldc, lse_i.data_ptr(),
                       _lib.ptr(ctx.rowgrad), n, _lib.ptr(g_mean_t), _lib.ptr(g_item_t), _lib.ptr(g_pos_t), _lib.ptr(g_neg_t),
                       _lib.ptr(dz1), n, _lib.ptr(dz2), n, _lib.ptr(dz3), n, 0,
                       ws.data_ptr(), ws.numel(), _lib.stream_ptr()), f"clica_{kind}_loss_bwd")
        return (dz1 if need1 else None), dz2, dz3, None, None


class RolledRows:
    """``torch.roll(source, shift, 0)`` that has not been computed: what a caller inside this package (the KITTI-masks solver) passes as
    ``z3_rec`` when the negatives are its own first views in another order.  ``LpSimCLRLoss`` with p >= 1 never reads the rolled copy
    (the row-wise log-sum-exp does not depend on the order of the negatives: `_PairLossSymFn`), so the contiguous copy and the roll
    launch of ``torch.roll`` on a strided view are not spent; every other consumer gets the real tensor from ``materialize()``."""

    def __init__(self, source, shift: int = 1):
        self.source, self.shift = source, int(shift)

    @property
    def shape(self):
        return self.source.shape

    def materialize(self):
        return torch.roll(self.source, self.shift, 0)


def _rolled_rows_of(z3, z1) -> bool:
    """Is `z3` the autograd result of ``torch.roll(z1, s, 0)`` (the reference's ``z3_rec = torch.roll(z1_rec, 1, 0)``,
    main_mlp.py:272)?  Read off the graph: z3's node is RollBackward0 along dim 0 and its input edge is z1's own gradient edge."""
    fn = getattr(z3, "grad_fn", None)
    if fn is None or z3.shape != z1.shape or fn.name() != "RollBackward0":
        return False
    try:
        if tuple(fn._saved_dims) != (0,):
            return False
        src, nr = fn.next_functions[0]
        if z1.grad_fn is not None:
            return src is z1.grad_fn and nr == z1.output_nr
        return z1.requires_grad and src is not None and getattr(src, "variable", None) is z1       # a leaf: its AccumulateGrad node
    except (AttributeError, IndexError):
        return False


def _scal(g):
    """0-dim upstream gradient -> fp32 device scalar the C ABI can read (None stays None)."""
    if g is None:
        return None
    if g.dtype == torch.float32 and g.is_cuda:
        return g.detach()                      # a 0-dim / 1-element tensor is its own contiguous buffer
    return g.detach().to(torch.float32).reshape(1).contiguous()


_WS_BYTES = {}       # (kind, B, B3, n, p, pow, compat/normalize) -> workspace bytes (a ctypes query per call otherwise)


def _lp_ws(desc, dev):
    key = (desc.B, desc.B3, desc.n, desc.p, desc.pow, desc.compat, desc.no_eps)
    nb = _WS_BYTES.get(key)
    if nb is None:
        fwd_b, bwd_b = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().clica_lp_loss_workspace_bytes(C.byref(desc), C.byref(fwd_b), C.byref(bwd_b)), "clica_lp_loss_workspace_bytes")
        nb = _WS_BYTES[key] = max(fwd_b.value, bwd_b.value)
    return _lib.workspace("lp_loss", nb, dev)


class _PairLossSymFn(torch.autograd.Function):
    """LpSimCLRLoss when the negatives ARE the anchors in another order (``z3_rec = roll(z1_rec)``): the row-wise log-sum-exp does
    not depend on the order of the negatives, so the forward reads z1 as the pool (no rolled copy), and the backward is ONE pair sweep
    (clica_lp_loss_bwd_sym: d_ij = d_ji, coefficient C_i w_ij + C_j w_ji) that delivers the complete d loss / d z1 -- the row part, the
    column part AND what autograd would have routed back through the roll.  What the training engine does, behind the reference's call."""

    @staticmethod
    def forward(ctx, z1, z2, desc):
        ctx.set_materialize_grads(False)      # (with materialised zeros for the unused per-item output the one-sweep backward was never taken)
        lib = _lib.load()
        (a, lda), (b, ldb) = _prep("z1_rec", z1), _prep("z2_con_z1_rec", z2)
        B = a.shape[0]
        out = torch.empty(3 * B + 3, dtype=torch.float32, device=a.device)
        loss_i, lse_i, means = out[:B], out[2 * B:3 * B], out[3 * B:]
        ws = _lp_ws(desc, a.device)
        p0 = out.data_ptr()
        _lib.check(lib.clica_lp_loss_fwd(C.byref(desc), a.data_ptr(), lda, b.data_ptr(), ldb, a.data_ptr(), lda,
                                         p0, p0 + 4 * B, p0 + 8 * B, p0 + 12 * B,
                                         None, 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "clica_lp_loss_fwd")
        ctx.save_for_backward(a, b, lse_i)
        ctx.lds, ctx.desc = (lda, ldb), desc
        mean, pos_mean, neg_mean = means.unbind(0)
        return mean, loss_i, pos_mean, neg_mean

    @staticmethod
    def backward(ctx, g_mean, g_item, g_pos, g_neg):
        lib = _lib.load()
        a, b, lse_i = ctx.saved_tensors
        (lda, ldb), desc, dev, n = ctx.lds, ctx.desc, a.device, a.shape[1]
        need1, need2 = ctx.needs_input_grad[:2]
        g_mean_t = _scal(g_mean)
        if g_mean_t is None:
            g_mean_t = torch.zeros(1, dtype=torch.float32, device=dev)
        g_pos_t, g_neg_t = _scal(g_pos), _scal(g_neg)
        # dz1 and dz2 as row blocks of ONE buffer: an encoder that produced z1 and z2 as one stacked call (cl_ica_amd/lazy.py) takes them
        # back as its stacked output gradient without a copy
        B1, B2 = a.shape[0], b.shape[0]
        dz = torch.empty((B1 + (B2 if need2 else 0), n), dtype=torch.float32, device=dev)
        dz1 = dz[:B1]
        dz2 = dz[B1:] if need2 else None
        ws = _lp_ws(desc, dev)
        PATHS["sym_one_sweep" if g_item is None and desc.p >= 1.0 else "sym_two_sweeps"] += 1
        if g_item is None and desc.p >= 1.0:
            _lib.check(lib.clica_lp_loss_bwd_sym(C.byref(desc), a.data_ptr(), lda, b.data_ptr(), ldb, a.data_ptr(), lda,
                                                 lse_i.data_ptr(), lse_i.data_ptr(), _lib.ptr(g_mean_t), _lib.ptr(g_pos_t), _lib.ptr(g_neg_t),
                                                 dz1.data_ptr(), n, _lib.ptr(dz2), n, ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                       "clica_lp_loss_bwd_sym")
        else:     # a per-item upstream gradient or p < 1: the generic two-sweep backward with z3 aliasing z1 (column part ADDED into dz1)
            g_item_t = None if g_item is None else g_item.detach().to(torch.float32).contiguous()
            _lib.check(lib.clica_lp_loss_bwd(C.byref(desc), a.data_ptr(), lda, b.data_ptr(), ldb, a.data_ptr(), lda, lse_i.data_ptr(),
                                             None, n, _lib.ptr(g_mean_t), _lib.ptr(g_item_t), _lib.ptr(g_pos_t), _lib.ptr(g_neg_t),
                                             dz1.data_ptr(), n, _lib.ptr(dz2), n, dz1.data_ptr(), n, 1,
                                             ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "clica_lp_loss_bwd")
        return (dz1 if need1 else None), dz2, None


SYM_ENABLED = True           # test hook (tests/test_gpu_next_rows.py): False = the generic two-sweep backward even where z3 = roll(z1) is recognised


def _sym_enabled() -> bool:
    return SYM_ENABLED


class LpSimCLRLoss(CLLoss):
    """Extended InfoNCE objective for non-normalized representations based on an Lp norm.

    Same arguments and defaults as the reference (losses.py:416-428):
        p: exponent of the norm; tau: temperature; alpha: weight between the two summands;
        simclr_compatibility_mode: logsumexp over [negatives, positive] instead of logmeanexp;
        pow: use the p-th power of the Lp norm instead of the norm.
    """

    def __init__(self, p: int, tau: float = 1.0, alpha: float = 0.5,
                 simclr_compatibility_mode: bool = False, pow: bool = True):
        self.p = p
        self.tau = tau
        self.alpha = alpha
        self.simclr_compatibility_mode = simclr_compatibility_mode
        self.pow = pow

    def _desc(self, B, B3, n):
        key = (B, B3, n, self.p, self.tau, self.alpha, self.simclr_compatibility_mode, self.pow)
        c = self.__dict__.get("_desc_cache")
        if c is None or c[0] != key:       # (the descriptor is read-only on the C side: one per shape and setting, not one per call)
            c = self._desc_cache = (key, _lib.LpLossDesc(B=B, B3=B3, n=n, p=float(self.p), tau=float(self.tau), alpha=float(self.alpha),
                                                         compat=int(bool(self.simclr_compatibility_mode)), pow=int(bool(self.pow))))
        return c[1]

    def loss(self, z1, z2_con_z1, z3, z1_rec, z2_con_z1_rec, z3_rec):
        del z1, z2_con_z1, z3   # unused by the reference as well (losses.py:431)
        # negatives that are the anchors in another order, not computed: RolledRows (callers inside this package) or the deferred
        # torch.roll of a deferred encoder output (lazy.LazyRoll: the reference's own z3_rec = torch.roll(z1_rec, 1, 0))
        rolled = isinstance(z3_rec, (RolledRows, lazy.LazyRoll)) and z3_rec.source is z1_rec
        if isinstance(z3_rec, (RolledRows, lazy.LazyRoll)) and not (rolled and _sym_enabled() and float(self.p) >= 1.0 and z1_rec.dim() == 2):
            z3_rec, rolled = z3_rec.materialize(), False
        z1_rec, z2_con_z1_rec = lazy.plain(z1_rec), lazy.plain(z2_con_z1_rec)
        z3_rec = z1_rec if rolled else lazy.plain(z3_rec)
        if z1_rec.shape != z2_con_z1_rec.shape or z1_rec.shape[1] != z3_rec.shape[1]:
            raise ValueError(f"shape mismatch: {tuple(z1_rec.shape)}, {tuple(z2_con_z1_rec.shape)}, {tuple(z3_rec.shape)}")
        desc = self._desc(z1_rec.shape[0], z3_rec.shape[0], z1_rec.shape[1])
        # the roll shortcut only for p >= 1: the p < 1 branch of the reference (losses.py:433-442) transposes the pair matrix, so row k
        # belongs to z3[k] = z1[k-1] and is combined with pos[k] -- reading z1 itself as the pool would pair pos[k] with the wrong row
        if rolled or (_sym_enabled() and float(self.p) >= 1.0 and z1_rec.dim() == 2 and _rolled_rows_of(z3_rec, z1_rec)):
            mean, per_item, pos_mean, neg_mean = _PairLossSymFn.apply(z1_rec, z2_con_z1_rec, desc)
        else:
            mean, per_item, pos_mean, neg_mean = _PairLossFn.apply(z1_rec, z2_con_z1_rec, z3_rec, "lp", desc)
        _share_items(_means_of(mean), (mean, pos_mean, neg_mean))
        return mean, per_item, [pos_mean, neg_mean]


class SimCLRLoss(CLLoss):
    """InfoNCE loss on dot-product similarities, optionally L2-normalised (losses.py:162-202)."""

    def __init__(self, normalize: bool = False, tau: float = 1.0, alpha: float = 0.5):
        self.normalize = normalize
        self.tau = tau
        self.alpha = alpha

    def loss(self, z1, z2_con_z1, z3, z1_rec, z2_con_z1_rec, z3_rec):
        del z1, z2_con_z1, z3
        z1_rec, z2_con_z1_rec, z3_rec = lazy.plain(z1_rec), lazy.plain(z2_con_z1_rec), lazy.plain(z3_rec)
        if z1_rec.shape != z2_con_z1_rec.shape or z1_rec.shape[1] != z3_rec.shape[1]:
            raise ValueError(f"shape mismatch: {tuple(z1_rec.shape)}, {tuple(z2_con_z1_rec.shape)}, {tuple(z3_rec.shape)}")
        desc = _lib.DotLossDesc(B=z1_rec.shape[0], B3=z3_rec.shape[0], n=z1_rec.shape[1], tau=float(self.tau),
                                alpha=float(self.alpha), normalize=int(bool(self.normalize)))
        mean, per_item, pos_mean, neg_mean = _PairLossFn.apply(z1_rec, z2_con_z1_rec, z3_rec, "dot", desc)
        _share_items(_means_of(mean), (mean, pos_mean, neg_mean))
        return mean, per_item, [pos_mean, neg_mean]


class ConditionalPairCLLoss(ABC):
    """Loss protocol with one positive pair (reference losses.py:32-46)."""

    @abstractmethod
    def loss(self, z1_rec, z2_con_z1_rec):
        ...

    def __call__(self, z1_rec, z2_con_z1_rec):
        return self.loss(z1_rec, z2_con_z1_rec)


class MarginalPairCLLoss(ABC):
    """Loss protocol with one negative pair (reference losses.py:49-63)."""

    @abstractmethod
    def loss(self, z1_rec, z3_rec):
        ...

    def __call__(self, z1_rec, z2_rec):
        return self.loss(z1_rec, z2_rec)


class UniformityLoss(MarginalPairCLLoss):
    """Loss over the negative pairs only (reference losses.py:205-222):
    ``per_item[j] = logmeanexp_i( -sum_k |z1_rec[i,k] - z3_rec[j,k]|^p )``, anchored at the rows of ``z3_rec``
    (the reference builds the pair tensor as ``z1.unsqueeze(0) - z3.unsqueeze(1)`` and reduces its last axis).
    Computed by the same tiled all-pairs kernel as LpSimCLRLoss: rows = z3_rec, pool = z1_rec, weight of the
    positive term 0, plain logmeanexp (no positive in the denominator), tau = 1; the kernel's per-row loss is
    2 * lse, hence the factor 0.5.  Returns ``(mean, per_item, [mean])``."""

    def __init__(self, p: int = 2.0):
        self.p = p

    def loss(self, z1_rec, z3_rec):
        z1_rec, z3_rec = lazy.plain(z1_rec), lazy.plain(z3_rec)
        if z1_rec.dim() != 2 or z3_rec.dim() != 2 or z1_rec.shape[1] != z3_rec.shape[1]:
            raise ValueError(f"shape mismatch: {tuple(z1_rec.shape)}, {tuple(z3_rec.shape)}")
        desc = _lib.LpLossDesc(B=z3_rec.shape[0], B3=z1_rec.shape[0], n=z1_rec.shape[1], p=float(self.p), tau=1.0,
                               alpha=0.0, compat=0, pow=1, no_eps=1)        # plain sum |d|^p for every p (no eps branch here)
        mean2, item2, _, _ = _PairLossFn.apply(z3_rec, z3_rec, z1_rec, "lp", desc)
        loss = 0.5 * mean2
        return loss, 0.5 * item2, [loss]


class AlignmentLoss(ConditionalPairCLLoss):
    """Loss over the positive pairs only (reference losses.py:225-241): ``per_item[i] = sum_k |z1_rec - z2_rec|^p``.
    Same kernel with the whole weight on the positive term (alpha = 1) and a one-row negatives pool (its
    log-sum-exp is finite and weighs 0).  Returns ``(mean, per_item, [mean])``."""

    def __init__(self, p: int = 2.0):
        self.p = p

    def loss(self, z1_rec, z2_rec):
        z1_rec, z2_rec = lazy.plain(z1_rec), lazy.plain(z2_rec)
        if z1_rec.dim() != 2 or z1_rec.shape != z2_rec.shape:
            raise ValueError(f"shape mismatch: {tuple(z1_rec.shape)}, {tuple(z2_rec.shape)}")
        desc = _lib.LpLossDesc(B=z1_rec.shape[0], B3=1, n=z1_rec.shape[1], p=float(self.p), tau=1.0, alpha=1.0, compat=0, pow=1,
                               no_eps=1)
        mean2, item2, _, _ = _PairLossFn.apply(z1_rec, z2_rec, z1_rec[:1].detach(), "lp", desc)
        loss = 0.5 * mean2
        return loss, 0.5 * item2, [loss]


class AlignmentUniformityLoss(CLLoss):
    """Convex combination of AlignmentLoss and UniformityLoss: ``(1 - alpha) * alignment(z1_rec, z2_con_z1_rec) + alpha *
    uniformity(z1_rec, z3_rec)`` (reference losses.py:242-250: same constructor, same weights ``[1 - alpha, alpha]``).  The reference
    builds it on CombinedCLLoss / SplitCombinedCLLoss, whose ``loss`` cannot be called with the pair losses' two-argument signature and
    raises (losses.py:118-151) -- this one is the working combination the docstring there describes, behind the 6-argument CLLoss call.
    Returns ``(mean, per_item, [alignment mean, uniformity mean])``; ``per_item`` is the same combination row by row when the two
    batches have the same number of rows (alignment is anchored at the rows of z1_rec, uniformity at those of z3_rec), else None."""

    def __init__(self, alpha=0.5, p=2.0):
        assert 0 <= alpha <= 1
        self.alpha, self.p = float(alpha), p
        self._align, self._unif = AlignmentLoss(p=p), UniformityLoss(p=p)

    def loss(self, z1, z2_con_z1, z3, z1_rec, z2_con_z1_rec, z3_rec):
        del z1, z2_con_z1, z3
        a, a_i, _ = self._align.loss(z1_rec, z2_con_z1_rec)
        u, u_i, _ = self._unif.loss(z1_rec, z3_rec)
        per_item = (1.0 - self.alpha) * a_i + self.alpha * u_i if a_i.shape == u_i.shape else None
        return (1.0 - self.alpha) * a + self.alpha * u, per_item, [a, u]
