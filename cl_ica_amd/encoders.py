"""MLP encoder factory with the reference's signature (/root/reference/encoders.py:10-85).

``get_mlp`` returns an ``nn.Sequential`` subclass holding exactly the modules the reference would
build (``nn.Linear`` / ``nn.LeakyReLU`` / head), so ``state_dict()`` keys and shapes
(``0.weight, 0.bias, 2.weight, ...``), ``.parameters()`` order, indexing (``f[-1].r``) and the
default initialisation are identical -- checkpoints load both ways.  ``forward`` does not call the
child modules: the Linear(+bias)(+LeakyReLU) stack runs as fused fp32-MFMA GEMM kernels
(cl_ica_amd/csrc/linear.hip) through one autograd node.
"""
from __future__ import annotations

from typing import List, Optional

import weakref

import torch
from torch import nn

from . import layers as ls
from . import lazy
from . import ops

__all__ = ["get_mlp", "FusedMLP", "NormedMLP"]


class _MLPStackFn(torch.autograd.Function):
    """Linear -> LeakyReLU(slope) -> ... -> Linear (no activation after the last)."""

    @staticmethod
    def forward(ctx, x, slope, *params):
        L = len(params) // 2
        acts = [x.detach()]
        for l in range(L):
            acts.append(ops.linear_fwd(acts[-1], params[2 * l], params[2 * l + 1], leaky=(l < L - 1), slope=slope))
        ctx.slope = slope
        ctx.L = L
        ctx.save_for_backward(*acts[:-1], *[p.detach() for p in params[0::2]])
        return acts[-1]

    @staticmethod
    def backward(ctx, gy):
        L, slope = ctx.L, ctx.slope
        saved = ctx.saved_tensors
        acts, Ws = saved[:L], saved[L:]
        grads: List[Optional[torch.Tensor]] = [None] * (2 * L)
        g = gy
        for l in reversed(range(L)):
            need_w, need_b = ctx.needs_input_grad[2 + 2 * l], ctx.needs_input_grad[3 + 2 * l]
            if need_w or need_b:
                dW, db = ops.linear_wgrad(g, acts[l], want_bias=need_b)
                grads[2 * l] = dW if need_w else None
                grads[2 * l + 1] = db if need_b else None
            if l > 0:
                # acts[l] is the LeakyReLU output feeding layer l: its sign gives act'
                g = ops.linear_dgrad(g, Ws[l], acts[l], slope)
            elif ctx.needs_input_grad[0]:
                g = ops.linear_dgrad(g, Ws[0], None, slope)
            else:
                g = None
        return (g, None, *grads)


class _MLPFusedFn(torch.autograd.Function):
    """The same stack on the WHOLE-ENCODER kernels the training engine uses (csrc/fused_mlp.hip, csrc/linear.hip): one
    launch for the forward (activation panel resident in LDS, sign bits of every hidden activation saved), one for the
    backward data chain, one grouped launch (+ one slab reduction) for every layer's dW / db.  Taken when every width is
    <= 512 (``ops.mlp_fwd_fusable``) and the batch is large enough to fill the chip (see ``_use_fused``)."""

    _pack_cache = {}      # device -> dict(key, shapes, packed, packed_t): fragment-order copies of the CURRENT weights
    _ws_cache = {}        # (device, rows, shapes) -> split-K slab workspace of the grouped weight-gradient launch (reused: every call
                          #   runs on the caller's stream, in order; allocating it per call was a 30 MB memset per backward)

    @staticmethod
    def _wgrad_ws(dev, M, shapes):
        # keyed by the stream as well: the slabs are scratch of ONE backward at a time, and backwards on different streams may overlap
        key = (dev, torch.cuda.current_stream(dev).cuda_stream, M, tuple(shapes))
        ws = _MLPFusedFn._ws_cache.get(key)
        if ws is None:
            if len(_MLPFusedFn._ws_cache) >= 4:      # each entry is 30-60 MB: a few shapes (train / eval batch), not a leak
                _MLPFusedFn._ws_cache.clear()
            ws = _MLPFusedFn._ws_cache[key] = ops.mlp_wgrad_workspace(M, shapes, dev)
        return ws

    @staticmethod
    def _weights_key(ws):
        return (ops.PARAM_EPOCH,) + tuple((w.data_ptr(), w._version, tuple(w.shape)) for w in ws)

    @staticmethod
    def _packed(params_w):
        """Fragment-order weight copies for the forward (``packed``) and the backward chain (``packed_t``), re-packed (one
        launch) only when a weight changed: the reference's train_step calls the encoder twice per step on the same weights."""
        key = _MLPFusedFn._weights_key(params_w)
        dev = params_w[0].device
        c = _MLPFusedFn._pack_cache.get(dev)
        if c is not None and c["key"] == key:
            return c["packed"], c["packed_t"], key
        shapes = [tuple(w.shape) for w in params_w]
        reuse = c is not None and c["shapes"] == shapes
        packed, packed_t = ops.mlp_pack_both([w.detach() for w in params_w], c["packed"] if reuse else None, c["packed_t"] if reuse else None)
        _MLPFusedFn._pack_cache[dev] = dict(key=key, shapes=shapes, packed=packed, packed_t=packed_t)
        return packed, packed_t, key

    @staticmethod
    def forward(ctx, x, slope, *params):
        L = len(params) // 2
        ws, bs = [p.detach() for p in params[0::2]], [p.detach() for p in params[1::2]]
        x = x.detach()
        M, dev = x.shape[0], x.device
        packed, packed_t, key = _MLPFusedFn._packed(params[0::2])
        outs = [torch.empty((M, w.shape[0]), dtype=torch.float32, device=dev) for w in ws]
        masks = ops.mlp_signmask_alloc(M, L - 1, dev, zero=False) + [None]
        ops.mlp_fwd(x, ws, bs, outs, slope, packed=packed, signmasks=masks)
        ctx.slope, ctx.L = slope, L
        ctx.pack_key = key
        ctx.params = params            # the leaf Parameters (for the in-place gradient accumulation of the flat optimizer, see backward)
        ctx.save_for_backward(x, *outs[:-1], *[m for m in masks[:-1]], *ws)
        return outs[-1]

    @staticmethod
    def backward(ctx, gy):
        L, slope = ctx.L, ctx.slope
        saved = ctx.saved_tensors
        x, acts, masks, ws = saved[0], list(saved[1:L]), list(saved[L:2 * L - 1]), list(saved[2 * L - 1:])
        gy = gy.contiguous()
        M, dev = gy.shape[0], gy.device
        # the fragment-order transposed weights must still describe the weights of the forward (they do unless a parameter was
        # written between forward and backward, which autograd itself would reject)
        cur = _MLPFusedFn._pack_cache.get(dev)
        if cur is not None and cur["key"] == ctx.pack_key:
            packed_t = cur["packed_t"]
        else:
            _, packed_t = ops.mlp_pack_both(ws)          # another model's forward re-packed the cache in between: rebuild
        chain = list(range(L - 1, 0, -1))
        dz = [torch.empty((M, ws[l].shape[1]), dtype=torch.float32, device=dev) for l in chain]     # dZ of layer l-1
        ops.mlp_dgrad_chain(gy, [ws[l] for l in chain], packed_t, [acts[l - 1] for l in chain], dz, slope,
                            masks_chain=[masks[l - 1] for l in chain])
        dz_of = {l - 1: dz[j] for j, l in enumerate(chain)}
        dz_of[L - 1] = gy
        need = ctx.needs_input_grad
        # cl_ica_amd.optim.Adam keeps every .grad as a view into its (zeroed) gradient arena: add dW / db into those views here
        # (one grouped launch, accumulate) and return None -- autograd then neither allocates fourteen gradient tensors per
        # encoder call nor launches an AccumulateGrad add per parameter (what Megatron's main_grad accumulation does)
        prm = ctx.params

        in_place = _inplace_ok(prm, need)
        wws = _MLPFusedFn._wgrad_ws(dev, M, [tuple(w.shape) for w in ws])
        if in_place:
            ops.mlp_wgrad([dz_of[l] for l in range(L)], [acts[l - 1] if l > 0 else x for l in range(L)],
                          [prm[2 * l].grad for l in range(L)], [prm[2 * l + 1].grad for l in range(L)], ws=wws, accumulate=True)
            grads = [None] * (2 * L)
        else:
            dWs = [torch.empty_like(w) for w in ws]
            dbs = [torch.empty(w.shape[0], dtype=torch.float32, device=dev) if need[3 + 2 * l] else None for l, w in enumerate(ws)]
            ops.mlp_wgrad([dz_of[l] for l in range(L)], [acts[l - 1] if l > 0 else x for l in range(L)], dWs, dbs, ws=wws)
            grads = []
            for l in range(L):
                grads += [dWs[l] if need[2 + 2 * l] else None, dbs[l]]
        dx = ops.linear_dgrad(dz_of[0], ws[0], None, slope) if need[0] else None
        return (dx, None, *grads)


def _adjacent_rows(gs, rows, n):
    """The tensors `gs` ([rows[i], n] each) as ONE [sum(rows), n] tensor if they already lie behind one another in one storage
    (contiguous fp32 row blocks), else None."""
    g0 = gs[0]
    if g0 is None or g0.dtype != torch.float32 or not g0.is_contiguous():
        return None
    st, off = g0.untyped_storage(), g0.storage_offset()
    ptr = st.data_ptr()
    for g, r in zip(gs, rows):
        if (g is None or g.dtype != torch.float32 or g.shape != (r, n) or not g.is_contiguous() or g.storage_offset() != off
                or g.untyped_storage().data_ptr() != ptr):
            return None
        off += r * n
    return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, g0.storage_offset(), (sum(rows), n), (n, 1))


class _SplitPlan:
    """What one (device, rows, layer shapes) call of the split whole-stack kernels needs and that does not change from step to step:
    which layers take planes / fp32 (`clica_mlp_wgrad_split_kind`), the byte layout of ONE buffer for the forward's saved tensors
    (planes, fp32 activations of tiny-dimension layers, sign bits) and of ONE for the backward's (dZ planes / fp32), and the constant
    ctypes argument arrays.  Per call that leaves two allocations and a handful of pointer sums where the first version made ~25
    tensors and ~40 list comprehensions over them (tools/dropin_hosttime.py: 104 -> ~35 us of host time in front of the forward launch,
    in a step whose device waits for exactly that)."""

    _cache = {}

    @staticmethod
    def get(dev, M, shapes, f16=False):
        key = (dev, M, shapes, f16)
        pl = _SplitPlan._cache.get(key)
        if pl is None:
            if len(_SplitPlan._cache) >= 16:
                _SplitPlan._cache.clear()
            pl = _SplitPlan._cache[key] = _SplitPlan(dev, M, shapes, f16)
        return pl

    def __init__(self, dev, M, shapes, f16=False):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        L = len(shapes)
        self.dev, self.M, self.L, self.shapes, self.f16 = dev, M, L, shapes, f16
        self.a_index = (C.c_int32 * L)(*range(L))                       # positions of layer l's operands in the f16x2 state's scale arrays
        self.d_index = (C.c_int32 * L)(*[L - 1 - l for l in range(L)])
        self.kinds = kinds = [ops.mlp_wgrad_split_kind(N, K) for N, K in shapes]        # 0: matrix-core weight gradient from planes
        nb = C.c_size_t()

        def planes_bytes(width, ones):
            fn = lib.clica_mlp_planes16_bytes if f16 else lib.clica_mlp_planes_bytes       # f16x2: 4 B per element, bf16x3: 6
            _lib.check(fn(M, width, 1 if ones else 0, C.byref(nb)), "clica_mlp_planes_bytes")
            return nb.value
        _lib.check(lib.clica_mlp_signmask_bytes(M, C.byref(nb)), "clica_mlp_signmask_bytes")
        mask_bytes = nb.value
        off = [0]

        def take(nbytes):
            o = off[0]
            off[0] = (o + nbytes + 255) & ~255
            return o
        # forward: layer l's output as planes (with the ones column) if the NEXT layer's weight gradient reads planes, as fp32 if it
        # reads fp32 (tiny-dimension layer); the last layer's output is the result (its own tensor)
        self.f_planes = [take(planes_bytes(shapes[l][0], True)) if (l + 1 < L and kinds[l + 1] == 0) else None for l in range(L)]
        self.f_outs = [take(4 * M * shapes[l][0]) if (l + 1 < L and kinds[l + 1] == 1) else None for l in range(L)]
        self.f_masks = [take(mask_bytes) if l + 1 < L else None for l in range(L)]
        self.f_bytes = max(off[0], 256)
        self.f_ldo = (C.c_int64 * L)(*[shapes[l][0] if (self.f_outs[l] is not None or l == L - 1) else 0 for l in range(L)])
        self.N = (C.c_int32 * L)(*[s[0] for s in shapes])
        self.K = (C.c_int32 * L)(*[s[1] for s in shapes])
        # backward chain (layer L-1 first, down to layer 1): link i differentiates layer chain[i] and emits dZ of layer chain[i] - 1
        self.chain = chain = list(range(L - 1, 0, -1))
        off[0] = 0
        self.b_planes = {j: (take(planes_bytes(shapes[j][0], False)) if kinds[j] == 0 else None) for j in range(L - 1)}
        self.b_f32 = {j: (take(4 * M * shapes[j][0]) if kinds[j] == 1 else None) for j in range(L - 1)}
        self.b_bytes = max(off[0], 256)
        n = len(chain)
        self.cN = (C.c_int32 * max(n, 1))(*[shapes[l][1] for l in chain])
        self.cK = (C.c_int32 * max(n, 1))(*[shapes[l][0] for l in chain])
        self.VP, self.VPc, self.I64, self.I64c = C.c_void_p * L, C.c_void_p * max(n, 1), C.c_int64 * L, C.c_int64 * max(n, 1)
        self.lddw = (C.c_int64 * L)(*[s[1] for s in shapes])


def _tail_ok(pl, shapes, kinds) -> bool:
    """May the backward chain produce the n-wide first / last layer's weight-gradient slabs itself (cached on the plan)?"""
    ok = bool(len(shapes) >= 3 and kinds[0] == 1 and kinds[-1] == 1 and all(k == 0 for k in kinds[1:-1]) and
              pl.f_outs[len(shapes) - 2] is not None and ops.mlp_chain_tail_supported(shapes))
    pl.tail_ok = ok
    return ok


class _MLPFusedSplitFn(torch.autograd.Function):
    """``_MLPFusedFn`` in the split-bf16 arithmetic the training engine uses by default (fp32 results from six bf16 products of
    exact 3-way operand splits: csrc/fused_mlp.hip mlp_split_k, csrc/wgrad_split.hip): forward stack, backward data chain and the
    grouped weight gradients, with hidden activations / dZ handed from kernel to kernel as bf16 planes where a matrix-core weight
    gradient is their only reader.  ``CLICA_SPLIT_BF16=0`` / ``CLICA_DROPIN_SPLIT=0`` keep the fp32-MFMA kernels.

    ``apply(slope, n_in, x_0 .. x_{n_in-1}, W_0, b_0, ...)``: the `n_in` row batches run STACKED through one launch per phase and
    come back as `n_in` outputs (row slices of one result) -- what `lazy.defer` does with the reference's two encoder calls per
    step, without CatBackward / SliceBackward nodes around the function (their backward was five small launches)."""

    _pack_cache = {}
    _ws_cache = {}

    @staticmethod
    def _packed(params_w, key=None, s16=None, force=False):
        """`s16`: the encoder's f16x2 context (`_S16Ctx`) or None for the bf16x3 arithmetic -- the two pack formats differ."""
        if key is None:
            key = _MLPFusedFn._weights_key(params_w)
        dev = params_w[0].device
        c = _MLPFusedSplitFn._pack_cache.get(dev)
        if c is not None and c["key"] == key and c["s16"] is s16 and not force:
            return c["packed"], c["packed_t"], key
        ptrs = tuple(k[0] for k in key[1:])
        if c is not None and c.get("ptrs") == ptrs and c["strides"] == [w.stride() for w in params_w] and c["s16"] is s16:
            # the same parameter tensors with new values (every training step): the argument arrays of the first call still hold
            from . import _lib
            if s16 is None:
                _lib.check(_lib.load().clica_mlp_pack_split_both(*c["args"], c["packed"].data_ptr(), c["packed_t"].data_ptr(), _lib.stream_ptr()),
                           "clica_mlp_pack_split_both")
            else:
                _lib.check(_lib.load().clica_mlp_pack_split16_both(*c["args"], c["packed"].data_ptr(), c["packed_t"].data_ptr(),
                                                                   s16.state.buf.data_ptr(), _lib.stream_ptr()), "clica_mlp_pack_split16_both")
            c["key"] = key
            return c["packed"], c["packed_t"], key
        import ctypes as C
        shapes = [tuple(w.shape) for w in params_w]
        reuse = c is not None and c["shapes"] == shapes and c["s16"] is s16
        ws = [w.detach() for w in params_w]
        packed, packed_t = ops.mlp_pack_split_both(ws, c["packed"] if reuse else None, c["packed_t"] if reuse else None,
                                                   state=None if s16 is None else s16.state)
        L = len(ws)
        args = None
        if all(w.dim() == 2 and w.stride(1) == 1 and w.is_cuda and w.dtype == torch.float32 for w in ws):
            args = (L, (C.c_void_p * L)(*[w.data_ptr() for w in ws]), (C.c_int64 * L)(*[w.stride(0) for w in ws]),
                    (C.c_int32 * L)(*[s_[0] for s_ in shapes]), (C.c_int32 * L)(*[s_[1] for s_ in shapes]))
        _MLPFusedSplitFn._pack_cache[dev] = dict(key=key, shapes=shapes, packed=packed, packed_t=packed_t, s16=s16,
                                                 params=[weakref.ref(w) for w in params_w], args=args,
                                                 ptrs=ptrs if args is not None else None, strides=[w.stride() for w in params_w])
        return packed, packed_t, key

    @staticmethod
    def repack_if_changed(*_a, **_k):
        """Optimizer post-step hook: bring the fragment-order weight copies up to date NOW, at the end of the step -- work that only
        depends on the new parameters, out of the way of the next step's first launch."""
        for dev, c in list(_MLPFusedSplitFn._pack_cache.items()):
            ws = [r() for r in c.get("params", ())]
            if ws and all(w is not None for w in ws):
                key = _MLPFusedFn._weights_key(ws)
                if key != c["key"]:
                    with torch.cuda.device(dev):
                        _MLPFusedSplitFn._packed(ws, key, s16=c["s16"])

    @staticmethod
    def _wgrad_ws(dev, M, shapes):
        from . import _lib
        key = (dev, _lib.stream_ptr(), M, tuple(shapes))
        ws = _MLPFusedSplitFn._ws_cache.get(key)
        if ws is None:
            if len(_MLPFusedSplitFn._ws_cache) >= 4:
                _MLPFusedSplitFn._ws_cache.clear()
            ws = _MLPFusedSplitFn._ws_cache[key] = ops.mlp_wgrad_split_workspace(M, shapes, dev)
        return ws

    @staticmethod
    def forward(ctx, slope, n_in, *args):
        from . import _lib
        xs, params = args[:n_in], args[n_in:]
        L = len(params) // 2
        x = xs[0].detach() if n_in == 1 else torch.cat([t.detach() for t in xs], 0)
        (x, ldx) = ops._mat("x", x)
        M, dev = x.shape[0], x.device
        pw = params[0::2]
        s16 = _s16_ctx(params, slope)            # f16x2 (the engine's default arithmetic) when the flat Adam owns these parameters, else bf16x3
        calibrate = s16 is not None and s16.begin_forward(pw)
        packed, _, key = _MLPFusedSplitFn._packed(pw, s16=s16, force=calibrate)
        pl = _SplitPlan.get(dev, M, tuple(tuple(w.shape) for w in pw), s16 is not None)
        arena = torch.empty(pl.f_bytes, dtype=torch.uint8, device=dev)
        y = torch.empty((M, pl.shapes[-1][0]), dtype=torch.float32, device=dev)
        base, VP = arena.data_ptr(), pl.VP
        bias = VP(*[(b if b.is_contiguous() else b.contiguous()).data_ptr() for b in params[1::2]])
        outs = VP(*[None if o is None else base + o for o in pl.f_outs[:-1]], y.data_ptr())
        fargs = (x.data_ptr(), ldx, M, None, 0, 0.0, None, 0, L, bias, outs, pl.f_ldo, pl.N, pl.K,
                 packed.data_ptr(), VP(*[None if o is None else base + o for o in pl.f_masks]),
                 VP(*[None if o is None else base + o for o in pl.f_planes]), float(slope))
        if s16 is None:
            _lib.check(_lib.load().clica_mlp_fwd_split(*fargs, _lib.stream_ptr()), "clica_mlp_fwd_split")
        else:
            st_ptr = s16.state.buf.data_ptr()
            if calibrate:
                # scales not measured on these parameters yet: this pass (weights packed and activations cut on the scales in force)
                # records every tensor's fp32 maximum, the update turns them into scales, weights are packed again -- then the real pass
                _lib.check(_lib.load().clica_mlp_fwd_split16(*fargs, st_ptr, _lib.stream_ptr()), "clica_mlp_fwd_split16")
                s16.state.update()
                _MLPFusedSplitFn._packed(pw, s16=s16, force=True)
                s16.forward_calibrated()
            _lib.check(_lib.load().clica_mlp_fwd_split16(*fargs, st_ptr, _lib.stream_ptr()), "clica_mlp_fwd_split16")
        ctx.s16 = s16
        ctx.slope, ctx.L, ctx.plan, ctx.n_in = slope, L, pl, n_in
        ctx.rows = [t.shape[0] for t in xs]
        ctx.pack_key = key
        ctx.params = params
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, arena, *[w.detach() for w in pw])
        if n_in == 1:
            return y
        # the row blocks as tensors of their own on y's storage (not autograd views of y: a view output of a multi-output function
        # may not be modified in place, an ordinary output may)
        res, off, N, st = [], 0, y.shape[1], y.untyped_storage()
        for r in ctx.rows:
            res.append(torch.empty(0, dtype=torch.float32, device=dev).set_(st, off * N, (r, N), (N, 1)))
            off += r
        return tuple(res)

    @staticmethod
    def backward(ctx, *gys):
        from . import _lib
        import ctypes as C
        L, slope, pl = ctx.L, ctx.slope, ctx.plan
        kinds, shapes, M, dev = pl.kinds, pl.shapes, pl.M, pl.dev
        sv = ctx.saved_tensors
        x, arena, ws = sv[0], sv[1], list(sv[2:])
        if ctx.n_in == 1:
            gy = gys[0]
            if gy is None:
                return (None,) * (3 + 2 * L)
            gy = gy.contiguous()
        else:
            if all(g is None for g in gys):
                return (None,) * (2 + ctx.n_in + 2 * L)
            gy = _adjacent_rows(gys, ctx.rows, shapes[-1][0])      # (the symmetric loss backward writes dz1 / dz2 into one buffer)
            if gy is None:
                gy = torch.cat([g if g is not None else x.new_zeros((r, shapes[-1][0])) for g, r in zip(gys, ctx.rows)], 0)
        _lib.require_cuda(gy, "grad_output")
        need = ctx.needs_input_grad
        need_x = any(need[2:2 + ctx.n_in])
        s16 = ctx.s16
        st = None if s16 is None else s16.state
        cur = _MLPFusedSplitFn._pack_cache.get(dev)
        if cur is not None and cur["key"] == ctx.pack_key and cur["s16"] is s16:
            packed_t = cur["packed_t"]
        else:
            _, packed_t = ops.mlp_pack_split_both(ws, state=st)
        lib, sp = _lib.load(), _lib.stream_ptr()
        fb = arena.data_ptr()
        barena = torch.empty(pl.b_bytes, dtype=torch.uint8, device=dev)
        bb = barena.data_ptr()
        # dZ of layer j: planes for a matrix-core weight gradient, fp32 for a tiny-dimension one (and for d loss / d input)
        dz0 = None
        f32_ptr = {j: (None if pl.b_f32[j] is None else bb + pl.b_f32[j]) for j in range(L - 1)}
        if L > 1 and need_x and f32_ptr[0] is None:
            dz0 = torch.empty((M, shapes[0][0]), dtype=torch.float32, device=dev)
            f32_ptr[0] = dz0.data_ptr()
        gy_planes = None
        use_tail = False
        if L > 1:
            chain, VPc = pl.chain, pl.VPc
            cargs = (gy.data_ptr(), gy.stride(0), M, L - 1, pl.cN, pl.cK, packed_t.data_ptr(),
                     VPc(*[fb + pl.f_masks[l - 1] for l in chain]),
                     VPc(*[f32_ptr[l - 1] for l in chain]),
                     pl.I64c(*[0 if f32_ptr[l - 1] is None else shapes[l - 1][0] for l in chain]),
                     VPc(*[None if pl.b_planes[l - 1] is None else bb + pl.b_planes[l - 1] for l in chain]),
                     float(slope))
            if st is None:
                _lib.check(lib.clica_mlp_dgrad_split(*cargs, sp), "clica_mlp_dgrad_split")
            else:
                if s16.begin_backward():
                    # gradient scales not measured yet: a link cuts its input on the scale in force and measures its OUTPUT in fp32, so
                    # every un-applied pass + update settles (at least) one more link of the chain (engine.calibrate_scales does the same)
                    for _ in range(L):
                        if kinds[L - 1] == 0:
                            gy_planes = ops.mlp_planes_from_f32(gy, False, out=gy_planes, state=st, tensor=(1, 0))
                        _lib.check(lib.clica_mlp_dgrad_split16(*cargs, st.buf.data_ptr(), sp), "clica_mlp_dgrad_split16")
                        st.update()
                    st.clear_flags()          # (passes on unmeasured scales: whatever they flagged is not a finding)
                    s16.backward_calibrated()
                # the chain also leaves the weight-gradient slabs of the n-wide first / last layer (clica_mlp_dgrad_split_tail: two small
                # fp32 products over each workgroup's 48 rows behind the last link) -- no tiny-dimension launch in front of the grouped GEMM
                use_tail = pl.tail_ok if hasattr(pl, "tail_ok") else _tail_ok(pl, shapes, kinds)
                if use_tail:
                    wsb_t = _MLPFusedSplitFn._wgrad_ws(dev, M, shapes)
                    tail = _lib.ChainTail(a_last=fb + pl.f_outs[L - 2], lda=shapes[L - 2][0], x=x.data_ptr(), ldx=x.stride(0), n_layers=L,
                                          N=pl.N, K=pl.K, wgrad_workspace=wsb_t.data_ptr(), wgrad_workspace_bytes=wsb_t.numel(), dy_parts=None)
                    _lib.check(lib.clica_mlp_dgrad_split_tail(*cargs, st.buf.data_ptr(), C.byref(tail), sp), "clica_mlp_dgrad_split_tail")
                else:
                    _lib.check(lib.clica_mlp_dgrad_split16(*cargs, st.buf.data_ptr(), sp), "clica_mlp_dgrad_split16")
        keep = [barena]
        # operands of the weight gradients, per layer: planes (kind 0) or fp32 (kind 1)
        dzp, xp, dzf, lddz, xf, ldxf = [None] * L, [None] * L, [None] * L, [0] * L, [None] * L, [0] * L
        for l in range(L):
            if kinds[l] == 0:
                if l == L - 1:
                    t = ops.mlp_planes_from_f32(gy, False, out=gy_planes, state=st, tensor=(1, 0)); keep.append(t); dzp[l] = t.data_ptr()
                else:
                    dzp[l] = bb + pl.b_planes[l]
                if l == 0:
                    t = ops.mlp_planes_from_f32(x, True, state=st, tensor=(0, 0)); keep.append(t); xp[l] = t.data_ptr()
                else:
                    xp[l] = fb + pl.f_planes[l - 1]
            else:
                dzf[l], lddz[l] = (gy.data_ptr(), gy.stride(0)) if l == L - 1 else (f32_ptr[l], shapes[l][0])
                xf[l], ldxf[l] = (x.data_ptr(), x.stride(0)) if l == 0 else (fb + pl.f_outs[l - 1], shapes[l - 1][0])
        prm = ctx.params
        in_place = _inplace_ok(prm, (True, True) + tuple(need[2 + ctx.n_in:]))
        if in_place:
            dWs, dbs, acc = [prm[2 * l].grad for l in range(L)], [prm[2 * l + 1].grad for l in range(L)], 1
        else:
            dWs = [torch.empty_like(w) for w in ws]
            dbs = [torch.empty(w.shape[0], dtype=torch.float32, device=dev) for w in ws]
            acc = 0
        wsb = _MLPFusedSplitFn._wgrad_ws(dev, M, shapes)
        VP = pl.VP
        wargs = (M, L, VP(*dzp), VP(*xp), VP(*dzf), pl.I64(*lddz), VP(*xf), pl.I64(*ldxf),
                 VP(*[w.data_ptr() for w in dWs]), pl.I64(*[w.stride(0) for w in dWs]), VP(*[b.data_ptr() for b in dbs]), pl.N, pl.K, acc)
        if st is None:
            _lib.check(lib.clica_mlp_wgrad_split(*wargs, wsb.data_ptr(), wsb.numel(), sp), "clica_mlp_wgrad_split")
        elif L > 1 and use_tail:
            _lib.check(lib.clica_mlp_wgrad_split16_tail(*wargs, st.buf.data_ptr(), pl.a_index, pl.d_index, 1, wsb.data_ptr(), wsb.numel(), sp),
                       "clica_mlp_wgrad_split16_tail")
        else:
            _lib.check(lib.clica_mlp_wgrad_split16(*wargs, st.buf.data_ptr(), pl.a_index, pl.d_index, wsb.data_ptr(), wsb.numel(), sp),
                       "clica_mlp_wgrad_split16")
        grads = [None] * (2 * L)
        if not in_place:
            for l in range(L):
                grads[2 * l] = dWs[l] if need[2 + ctx.n_in + 2 * l] else None
                grads[2 * l + 1] = dbs[l] if need[3 + ctx.n_in + 2 * l] else None
        dxs = [None] * ctx.n_in
        if need_x:
            d0 = gy if L == 1 else (dz0 if dz0 is not None else torch.as_strided(barena.view(torch.float32), (M, shapes[0][0]), (shapes[0][0], 1), pl.b_f32[0] // 4))
            dx = ops.linear_dgrad(d0, ws[0], None, slope)
            off = 0
            for i, r in enumerate(ctx.rows):
                dxs[i] = dx[off:off + r] if need[2 + i] else None
                off += r
        del keep
        return (None, None, *dxs, *grads)


S16_ENABLED = True           # test hook: False keeps the drop-in encoder on the bf16x3 arithmetic (as CLICA_SPLIT_ARITH=bf16 does)


class _S16Ctx:
    """f16x2 arithmetic of ONE drop-in encoder (include/clica.h "f16x2 arithmetic"; the engine's default since round 5): the device state
    with the per-tensor scales, and what the host knows about them.  Scales follow the data with one step's delay, so they are MEASURED
    before the first step and after the parameters were written from outside (version counters): the forward runs one un-applied pass +
    update, the first backward L un-applied chain passes (`_MLPFusedSplitFn`).  From then on the update rides in the flat Adam's launch
    (cl_ica_amd/optim.py), which also honours THE GUARD: a step whose producers met a value beyond its scale leaves the parameters
    untouched (counted; `arith_state`)."""

    def __init__(self, n_layers, device, opt):
        self.state = ops.Split16(n_layers, device)
        self.opt = weakref.ref(opt)
        self.versions = None
        self.fwd_ok = self.bwd_ok = False

    def begin_forward(self, pw) -> bool:
        """True: this forward has to measure the scales first."""
        v = sum(w._version for w in pw)
        if v != self.versions:
            self.versions, self.fwd_ok, self.bwd_ok = v, False, False
        return not self.fwd_ok and not torch.cuda.is_current_stream_capturing()

    def forward_calibrated(self):
        self.fwd_ok = True

    def begin_backward(self) -> bool:
        return not self.bwd_ok and not torch.cuda.is_current_stream_capturing()

    def backward_calibrated(self):
        self.bwd_ok = True


def _s16_ctx(params, slope):
    """The f16x2 context of the encoder with these parameters, or None (bf16x3): needs `cl_ica_amd.optim.Adam` over ALL of them -- the
    launch that applies the step is what the arithmetic's guard acts through --, single rank, <= 8 layers, a LeakyReLU slope in (0, 1)."""
    import os
    if not S16_ENABLED or not (0.0 < slope < 1.0) or len(params) > 16 or os.environ.get("CLICA_SPLIT_ARITH", "f16").lower() != "f16":
        return None
    opt = None
    for q in params:
        r = q.__dict__.get("_clica_flat_opt")
        o = r() if r is not None else None
        if o is None or (opt is not None and o is not opt):
            return None
        opt = o
    if opt.world > 1:
        return None
    c = params[0].__dict__.get("_clica_s16")
    if c is None or c.opt() is not opt:
        c = _S16Ctx(len(params) // 2, params[0].device, opt)
        params[0].__dict__["_clica_s16"] = c
    return c if opt._bind_s16(c) else None


def arith_state(module) -> dict:
    """Which arithmetic the whole-encoder kernels of a drop-in encoder run in and, for f16x2, the state of its scales and of the guard
    (host read + sync: log points, tests): flags (bit 0 overflow seen, bit 1 a step was withheld), `skipped` = steps withheld so far."""
    lin = [m for m in module if isinstance(m, nn.Linear)]
    c = lin[0].weight.__dict__.get("_clica_s16") if lin else None
    if c is None:
        return dict(arith="bf16x3")
    return dict(arith="f16x2", **c.state.read(), **{k: v for k, v in c.state.guard().items() if k in ("skipped", "poisoned")})


def _after_step():
    _MLPFusedSplitFn.repack_if_changed()          # weights re-packed at step end (the next forward finds them ready)


lazy.AFTER_STEP.append(_after_step)


def _dropin_split(fits: bool) -> bool:
    """`fits`: the layers' padded widths fit the on-chip bias table of mlp_split_k (FusedMLP._structure)."""
    import os
    return fits and os.environ.get("CLICA_SPLIT_BF16", "1") != "0"


def _inplace_grads() -> bool:
    """In-place weight-gradient accumulation (`_inplace_ok`) is always on."""
    return True


def _inplace_ok(prm, need) -> bool:
    """May this backward add dW / db straight into the flat optimizer's `.grad` views and hand autograd None?

    Only when that is indistinguishable from what autograd itself would do with the returned gradients:
      * every parameter's `.grad` IS the view `cl_ica_amd.optim.Adam` installed into its gradient arena;
      * this backward pass really accumulates into the leaves -- i.e. it is `loss.backward()`, not
        `torch.autograd.grad(y, x)` / `autograd.functional.jacobian` (reference losses.py:279) / `backward(inputs=[...])`, in which
        the parameters' AccumulateGrad nodes do not run: `ctx.needs_input_grad` cannot tell (it reflects `requires_grad` at
        forward time), the engine's own execution plan can (`torch._C._will_engine_execute_node`);
      * no tensor hook / post-accumulate-grad hook is registered on a parameter (they would never fire).
    Otherwise the gradients are returned to autograd as ordinary tensors."""
    if not (_inplace_grads() and all(need[2:])):
        return False
    will_run = torch._C._will_engine_execute_node
    try:
        for q in prm:
            v, g = q.__dict__.get("_clica_grad_view"), q.grad
            if v is None or g is None or g.data_ptr() != v.data_ptr() or not g.is_contiguous():
                return False
            if q._backward_hooks or q.__dict__.get("_post_accumulate_grad_hooks"):
                return False
            node = q.__dict__.get("_clica_acc_node")
            if node is None:
                node = q._clica_acc_node = torch.autograd.graph.get_gradient_edge(q).node     # the leaf's AccumulateGrad node
            if not will_run(node):
                return False
    except RuntimeError:          # autograd.grad(loss, params): the engine CAPTURES the leaf gradients -- they must be returned
        return False
    return True


def _use_fused(fusable: bool, M: int) -> bool:
    """Whole-encoder kernels for the autograd path?  They own 48 rows per workgroup for the whole stack, so they want
    half of the 256 CUs busy (measured at 128 workgroups = one B = 6144 encoder call of the reference's train_step: 645 against
    594 steps/s through the per-layer GEMMs); smaller batches (and wide encoders) take the per-layer GEMMs.
    `FUSED_MODE` ("auto" / "0" / "1"; tests force either product path on one shape).  `fusable`: > 1 layers, all with bias, every width
    <= 512 (FusedMLP._structure)."""
    e = FUSED_MODE
    if not fusable or e == "0":
        return False
    return True if e == "1" else (M + 47) // 48 >= 128


FUSED_MODE = "auto"          # test hook (tests/test_gpu_mlp.py, test_gpu_next_rows.py): monkeypatch.setattr(encoders, "FUSED_MODE", "0" | "1")


class FusedMLP(nn.Sequential):
    """``nn.Sequential`` whose forward runs the fused HIP path (same modules, same state dict)."""

    def _structure(self):
        """(linears, slope, params, post modules, all parameters, fusable, split) of the current module list -- cached, re-derived when a
        child module or one of its parameters has been replaced (the walk over the children was 15 us per call, twice per step)."""
        c = self.__dict__.get("_clica_structure")
        if c is not None:
            mods = self._modules
            if len(mods) == c[7] and all(mods[k] is m for k, m in c[8]) and all(m._parameters[nm] is q for m, nm, q in c[9]):
                return c
        mods = list(self)
        linears = [m for m in mods if isinstance(m, nn.Linear)]
        slopes = {m.negative_slope for m in mods if isinstance(m, nn.LeakyReLU)}
        slope = slopes.pop() if slopes else 0.01
        params = []
        for lin in linears:
            params += [lin.weight, lin.bias]
        post = [m for m in mods if isinstance(m, (ls.RescaleLayer, ls.SoftclipLayer))]
        every = [p for p in params if p is not None]      # (not self.parameters(): a module-tree walk per call)
        for m in post:
            every += [q for q in m._parameters.values() if q is not None]
        fusable = len(linears) > 1 and all(lin.bias is not None for lin in linears) and ops.mlp_fwd_fusable([lin.weight for lin in linears])
        split = sum((lin.out_features + 31) // 32 * 32 for lin in linears) <= 3456        # on-chip bias table of mlp_split_k (fused_mlp.hip)
        guard = [(m, nm, q) for m in linears + post for nm, q in m._parameters.items()]
        c = (linears, slope, params, post, every, fusable, split, len(self._modules), list(self._modules.items()), guard)
        self.__dict__["_clica_structure"] = c
        return c

    def forward(self, x):
        x = lazy.plain(x)
        linears, slope, params, post, every, fusable, split = self._structure()[:7]
        if x.dim() != 2:
            x = x.reshape(-1, x.shape[-1])

        def compute(xx):
            if _use_fused(fusable, xx.shape[0]):
                y = _MLPFusedSplitFn.apply(slope, 1, xx, *params) if _dropin_split(split) else _MLPFusedFn.apply(xx, slope, *params)
            else:
                y = _MLPStackFn.apply(xx, slope, *params)
            for m in post:
                y = m(y)
            return y

        def compute_many(xlist):       # several pending calls as ONE function with several outputs (None: not on this path)
            if post or not (_use_fused(fusable, sum(t.shape[0] for t in xlist)) and _dropin_split(split)):
                return None
            return list(_MLPFusedSplitFn.apply(slope, len(xlist), *xlist, *params))
        # The reference's train_step calls the encoder twice per step (main_mlp.py:270-271).  A training call that does not fill the
        # chip on its own (fewer than 256 panels of 48 rows) is DEFERRED: if the same module is called again before anything uses the
        # result, both batches run as one stacked launch per phase (cl_ica_amd/lazy.py); any other use computes it right away.
        if (lazy.enabled() and torch.is_grad_enabled() and x.is_cuda and (x.shape[0] + 47) // 48 < 256
                and any(p.requires_grad for p in every)):
            return lazy.defer(self, x, compute, (x.shape[0], linears[-1].out_features), every, compute_many=compute_many)
        return compute(x)


class NormedMLP(nn.Sequential):
    """get_mlp(layer_normalization="bn" | "gn"): same modules and state dict as the reference's Sequential; Linear and
    LeakyReLU run on the HIP kernels (per layer), the normalisation modules are torch's."""

    def forward(self, x):
        x = lazy.plain(x)
        if x.dim() != 2:
            x = x.reshape(-1, x.shape[-1])
        for m in self:
            if isinstance(m, nn.Linear):
                x = _MLPStackFn.apply(x, 0.01, m.weight, m.bias)       # one-layer stack: no activation
            elif isinstance(m, nn.LeakyReLU):
                x = ls._LeakyFn.apply(x.contiguous(), float(m.negative_slope))
            else:
                x = m(x)
        return x


def get_mlp(n_in: int, n_out: int, layers: List[int], layer_normalization: Optional[str] = None,
            output_normalization: Optional[str] = None, output_normalization_kwargs=None):
    """Creates an MLP (same arguments as encoders.py:10-23).

    Args:
        n_in: dimensionality of the input data
        n_out: dimensionality of the output data
        layers: number of neurons for each hidden layer (the reference appends ``n_out`` to the
            caller's list in place, encoders.py:56; reproduced)
        layer_normalization: None | "bn" | "gn" (encoders.py:41-44).  With a normalisation between the layers the
            stack cannot run as one fused kernel: every Linear runs on the HIP GEMM kernels, the activation on the HIP
            elementwise kernel, and BatchNorm1d / GroupNorm(1, .) are torch's own device modules (none of the reference's
            drivers passes this argument, so it is outside the measured hot path)
        output_normalization: None | "fixed_sphere" | "learnable_sphere" | "fixed_box" | "learnable_box"
        output_normalization_kwargs: forwarded to the head (e.g. ``init_r`` for the sphere)
    """
    if layer_normalization not in (None, "bn", "gn"):
        raise ValueError("layer_normalization")
    if len(layers) == 0:
        raise ValueError("get_mlp needs at least one hidden layer (the reference's empty-layers "
                         "branch raises as well, encoders.py:54)")
    modules: List[nn.Module] = []
    layers.append(n_out)
    width = n_in
    for i, l in enumerate(layers):
        modules.append(nn.Linear(width, l))
        if i < len(layers) - 1:
            if layer_normalization == "bn":
                modules.append(nn.BatchNorm1d(l))
            elif layer_normalization == "gn":
                modules.append(nn.GroupNorm(1, l))
            modules.append(nn.LeakyReLU())
        width = l
    kw = output_normalization_kwargs or {}
    if output_normalization == "fixed_sphere":
        modules.append(ls.RescaleLayer(fixed_r=True, **kw))
    elif output_normalization == "learnable_sphere":
        modules.append(ls.RescaleLayer(init_r=1.0, fixed_r=False))
    elif output_normalization == "fixed_box":
        modules.append(ls.SoftclipLayer(n=n_out, fixed_abs_bound=True, **kw))
    elif output_normalization == "learnable_box":
        modules.append(ls.SoftclipLayer(n=n_out, fixed_abs_bound=False, **kw))
    elif output_normalization is not None:
        raise ValueError("output_normalization")
    return (NormedMLP if layer_normalization else FusedMLP)(*modules)
