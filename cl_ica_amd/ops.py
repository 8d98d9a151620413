"""Thin functional wrappers over the C ABI (include/clica.h).  Every function launches HIP
kernels on torch's current stream and returns torch tensors; there is no other code path."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, load, ptr, require_cuda, rowmajor, stream_ptr, workspace, ClicaError


def _mat(name: str, t: torch.Tensor) -> Tuple[torch.Tensor, int]:
    if t.dim() != 2:
        raise ValueError(f"{name} must be 2-D, got {tuple(t.shape)}")
    require_cuda(t, name)
    return rowmajor(t.detach())


# ------------------------------------------------------------------------------- Linear
def linear_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], leaky: bool,
               slope: float = 0.01, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Y = act(x W^T + b); act = LeakyReLU(slope) if `leaky` (encoders.py:38-48)."""
    (x, ldx), (w, ldw) = _mat("x", x), _mat("weight", weight)
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"weight {tuple(w.shape)} does not match input {tuple(x.shape)}")
    if bias is not None:
        require_cuda(bias, "bias")
        bias = bias.detach().contiguous()
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(load().clica_linear_fwd(x.data_ptr(), ldx, w.data_ptr(), ldw, ptr(bias), y.data_ptr(), y.stride(0),
                                  M, N, K, int(leaky), float(slope), stream_ptr()), "clica_linear_fwd")
    return y


def linear_dgrad(dy: torch.Tensor, weight: torch.Tensor, xact: Optional[torch.Tensor], slope: float = 0.01,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dX = (dY W) * act'(xact)  (xact = saved activation output feeding this layer, or None)."""
    (dy, lddy), (w, ldw) = _mat("dy", dy), _mat("weight", weight)
    M, N = dy.shape
    K = w.shape[1]
    xa, ldxa = (None, 0)
    if xact is not None:
        xa, ldxa = _mat("xact", xact)
    dx = out if out is not None else torch.empty((M, K), dtype=torch.float32, device=dy.device)
    check(load().clica_linear_dgrad(dy.data_ptr(), lddy, w.data_ptr(), ldw, ptr(xa), ldxa, float(slope),
                                    dx.data_ptr(), dx.stride(0), M, N, K, stream_ptr()), "clica_linear_dgrad")
    return dx


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, dW: Optional[torch.Tensor] = None,
                 db: Optional[torch.Tensor] = None, accumulate: bool = False, want_bias: bool = True,
                 ws: Optional[torch.Tensor] = None):
    """dW = dY^T X, db = column sums of dY."""
    (dy, lddy), (x, ldx) = _mat("dy", dy), _mat("x", x)
    M, N = dy.shape
    K = x.shape[1]
    if dW is None:
        dW = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    if db is None and want_bias:
        db = torch.empty((N,), dtype=torch.float32, device=dy.device)
    if ws is None:
        nbytes = C.c_size_t()
        check(load().clica_linear_wgrad_workspace_bytes(M, N, K, C.byref(nbytes)), "clica_linear_wgrad_workspace_bytes")
        ws = workspace("wgrad", nbytes.value, dy.device)
    check(load().clica_linear_wgrad(dy.data_ptr(), lddy, x.data_ptr(), ldx, dW.data_ptr(), dW.stride(0), ptr(db),
                                    M, N, K, int(accumulate), ws.data_ptr(), ws.numel(), stream_ptr()),
          "clica_linear_wgrad")
    return dW, db


def mlp_wgrad_workspace(M: int, shapes, device) -> torch.Tensor:
    """Workspace for `mlp_wgrad` over layers with weight shapes `shapes` = [(N_l, K_l), ...]."""
    n = len(shapes)
    I32 = C.c_int32 * n
    nbytes = C.c_size_t()
    check(load().clica_mlp_wgrad_workspace_bytes(int(M), n, I32(*[s[0] for s in shapes]), I32(*[s[1] for s in shapes]), C.byref(nbytes)),
          "clica_mlp_wgrad_workspace_bytes")
    return torch.zeros(nbytes.value, dtype=torch.uint8, device=device)


def mlp_wgrad(dzs, xs, dWs, dbs, ws: Optional[torch.Tensor] = None, accumulate: bool = False):
    """All layers' dW[l] = dZ[l]^T X[l], db[l] = column sums of dZ[l] in two launches (clica_mlp_wgrad)."""
    n = len(dzs)
    dz = [_mat(f"dz[{l}]", t) for l, t in enumerate(dzs)]
    xx = [_mat(f"x[{l}]", t) for l, t in enumerate(xs)]
    M = dz[0][0].shape[0]
    shapes = [(d.shape[1], x.shape[1]) for (d, _), (x, _) in zip(dz, xx)]
    if ws is None:
        ws = mlp_wgrad_workspace(M, shapes, dz[0][0].device)
    VP, I64, I32 = C.c_void_p * n, C.c_int64 * n, C.c_int32 * n
    check(load().clica_mlp_wgrad(M, n, VP(*[d.data_ptr() for d, _ in dz]), I64(*[ld for _, ld in dz]),
                                 VP(*[x.data_ptr() for x, _ in xx]), I64(*[ld for _, ld in xx]),
                                 VP(*[w.data_ptr() for w in dWs]), I64(*[w.stride(0) for w in dWs]),
                                 VP(*[ptr(b) for b in dbs]), I32(*[s[0] for s in shapes]), I32(*[s[1] for s in shapes]),
                                 int(accumulate), ws.data_ptr(), ws.numel(), stream_ptr()), "clica_mlp_wgrad")
    return dWs, dbs


MLP_FUSED_MAX_WIDTH = 512
MLP_FUSED_MAX_LAYERS = 8


def mlp_fwd_fusable(weights) -> bool:
    return len(weights) <= MLP_FUSED_MAX_LAYERS and all(max(w.shape) <= MLP_FUSED_MAX_WIDTH for w in weights)


def mlp_pack_weights(weights, packed: Optional[torch.Tensor] = None, transpose: bool = False) -> torch.Tensor:
    """Weights in MFMA fragment order for `mlp_fwd(..., packed=...)` (clica_mlp_pack), or -- with
    ``transpose`` and the layers given in chain order (last layer first) -- for `mlp_dgrad_chain`.
    Re-run after every parameter update; `packed` is reused when given."""
    L = len(weights)
    ws = [_mat(f"weight[{l}]", w) for l, w in enumerate(weights)]
    I32 = C.c_int32 * L
    Ns, Ks = I32(*[w.shape[0] for w, _ in ws]), I32(*[w.shape[1] for w, _ in ws])
    if packed is None:
        nb = C.c_size_t()
        check(load().clica_mlp_pack_bytes(L, Ns, Ks, int(transpose), C.byref(nb)), "clica_mlp_pack_bytes")
        packed = torch.zeros(nb.value // 4, dtype=torch.float32, device=ws[0][0].device)   # zero padding written once
    check(load().clica_mlp_pack(L, (C.c_void_p * L)(*[w.data_ptr() for w, _ in ws]), (C.c_int64 * L)(*[ld for _, ld in ws]),
                                Ns, Ks, int(transpose), packed.data_ptr(), stream_ptr()), "clica_mlp_pack")
    return packed


def mlp_pack_both(weights, packed: Optional[torch.Tensor] = None, packed_t: Optional[torch.Tensor] = None):
    """Both fragment-order copies a training step needs in ONE launch (clica_mlp_pack_both): `packed` for
    `mlp_fwd` (layers 0..L-1) and `packed_t` for `mlp_dgrad_chain` (layers L-1..1, transposed)."""
    L = len(weights)
    ws = [_mat(f"weight[{l}]", w) for l, w in enumerate(weights)]
    I32 = C.c_int32 * L
    Ns, Ks = I32(*[w.shape[0] for w, _ in ws]), I32(*[w.shape[1] for w, _ in ws])
    dev = ws[0][0].device
    if packed is None:
        nb = C.c_size_t()
        check(load().clica_mlp_pack_bytes(L, Ns, Ks, 0, C.byref(nb)), "clica_mlp_pack_bytes")
        packed = torch.zeros(nb.value // 4, dtype=torch.float32, device=dev)
    if packed_t is None:
        nb = C.c_size_t()
        chain = list(range(L - 1, 0, -1))
        I32c = C.c_int32 * (L - 1)
        check(load().clica_mlp_pack_bytes(L - 1, I32c(*[ws[l][0].shape[0] for l in chain]), I32c(*[ws[l][0].shape[1] for l in chain]), 1,
                                          C.byref(nb)), "clica_mlp_pack_bytes")
        packed_t = torch.zeros(nb.value // 4, dtype=torch.float32, device=dev)
    check(load().clica_mlp_pack_both(L, (C.c_void_p * L)(*[w.data_ptr() for w, _ in ws]), (C.c_int64 * L)(*[ld for _, ld in ws]),
                                     Ns, Ks, packed.data_ptr(), packed_t.data_ptr(), stream_ptr()), "clica_mlp_pack_both")
    return packed, packed_t


# ---- opt-in split-bf16 arithmetic (clica_mlp_*_split): fp32-grade results on the bf16 matrix cores -------------------
def mlp_pack_split_both(weights, packed: Optional[torch.Tensor] = None, packed_t: Optional[torch.Tensor] = None, state=None):
    """Fragment-order piece copies for `mlp_fwd_split` (layers 0..L-1) and `mlp_dgrad_chain_split` (layers L-1..1, transposed) in one
    launch: bf16x3, or -- with a `Split16` state -- f16x2 scaled by the state's weight scales (clica_mlp_pack_split16_both)."""
    L = len(weights)
    ws = [_mat(f"weight[{l}]", w) for l, w in enumerate(weights)]
    I32 = C.c_int32 * L
    Ns, Ks = I32(*[w.shape[0] for w, _ in ws]), I32(*[w.shape[1] for w, _ in ws])
    dev = ws[0][0].device
    nb = C.c_size_t()
    nbytes = load().clica_mlp_pack_split_bytes if state is None else load().clica_mlp_pack_split16_bytes
    if packed is None:
        check(nbytes(L, Ns, Ks, 0, C.byref(nb)), "clica_mlp_pack_split_bytes")
        packed = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
    if packed_t is None:
        chain = list(range(L - 1, 0, -1))
        I32c = C.c_int32 * (L - 1)
        check(nbytes(L - 1, I32c(*[ws[l][0].shape[0] for l in chain]), I32c(*[ws[l][0].shape[1] for l in chain]), 1, C.byref(nb)),
              "clica_mlp_pack_split_bytes")
        packed_t = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
    Wp, ldp = (C.c_void_p * L)(*[w.data_ptr() for w, _ in ws]), (C.c_int64 * L)(*[ld for _, ld in ws])
    if state is None:
        check(load().clica_mlp_pack_split_both(L, Wp, ldp, Ns, Ks, packed.data_ptr(), packed_t.data_ptr(), stream_ptr()), "clica_mlp_pack_split_both")
    else:
        check(load().clica_mlp_pack_split16_both(L, Wp, ldp, Ns, Ks, packed.data_ptr(), packed_t.data_ptr(), state.buf.data_ptr(), stream_ptr()),
              "clica_mlp_pack_split16_both")
    return packed, packed_t


class Split16:
    """Device state of the f16x2 encoder arithmetic of ONE encoder (include/clica.h, "f16x2 arithmetic"): per-tensor scales in force and
    the running maxima the producer kernels record.  `update()` (one tiny launch, once per training step after its last producer) turns
    the maxima into the next step's scales; `read()` synchronises and reports flags / scales (log points, tests)."""

    def __init__(self, n_layers: int, device):
        nb = C.c_size_t()
        check(load().clica_split16_state_bytes(C.byref(nb)), "clica_split16_state_bytes")
        self.n_layers = int(n_layers)
        self.buf = torch.zeros(nb.value, dtype=torch.uint8, device=device)
        check(load().clica_split16_state_init(self.buf.data_ptr(), stream_ptr()), "clica_split16_state_init")

    def update(self):
        check(load().clica_split16_update(self.buf.data_ptr(), self.n_layers, stream_ptr()), "clica_split16_update")

    def clear_flags(self):
        check(load().clica_split16_clear_flags(self.buf.data_ptr(), stream_ptr()), "clica_split16_clear_flags")

    def guard(self) -> dict:
        """State of the device-side guard (include/clica.h, "THE GUARD of the f16x2 arithmetic"): sticky flags, steps withheld so far, scale
        updates so far, and whether the step whose producers ran last is poisoned.  Host read + sync: log points, tests."""
        fl, sk, up, po = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(load().clica_split16_guard(self.buf.data_ptr(), C.byref(fl), C.byref(sk), C.byref(up), C.byref(po), stream_ptr()), "clica_split16_guard")
        return dict(flags=int(fl.value), skipped=int(sk.value), updates=int(up.value), poisoned=bool(po.value))

    def set_dp_poison(self, slot: Optional[torch.Tensor]):
        """Data parallel: the optimizer launches take the step's verdict from the device float `slot` (all-reduced by the caller)."""
        check(load().clica_split16_set_dp_poison(self.buf.data_ptr(), ptr(slot), stream_ptr()), "clica_split16_set_dp_poison")
        self._dp_slot = slot

    def poison_export(self, slot: torch.Tensor):
        check(load().clica_split16_poison_export(self.buf.data_ptr(), slot.data_ptr(), stream_ptr()), "clica_split16_poison_export")

    def read(self) -> dict:
        fl, up = C.c_int32(), C.c_int32()
        a, d, w, pa, pd = (C.c_float * 9)(), (C.c_float * 9)(), (C.c_float * 9)(), (C.c_float * 9)(), (C.c_float * 9)()
        check(load().clica_split16_read(self.buf.data_ptr(), C.byref(fl), C.byref(up), a, d, w, pa, pd, stream_ptr()), "clica_split16_read")
        L = self.n_layers
        return dict(flags=int(fl.value), updates=int(up.value), scales_a=list(a)[:L + 1], scales_d=list(d)[:L], scales_w=list(w)[:L],
                    last_scales_a=list(pa)[:L + 1], last_scales_d=list(pd)[:L])


_PLANES_BYTES = {}


def mlp_planes_alloc(M: int, width: int, ones: bool, device, zero: bool = True, f16: bool = False) -> torch.Tensor:
    """Opaque buffer for the bf16-plane copy of an [M, width] layer output (operand format of `mlp_wgrad_split`).
    `zero=False`: uninitialised -- for buffers a producer kernel writes completely (the whole-stack kernels and
    `mlp_planes_from_f32` write every piece of every 16-row group, padding and ones column included; pinned by the NaN-fill in
    tests/test_gpu_mlp.py::test_split_bf16_wgrad_matches_fp64).  The drop-in path allocates per call: zeroing was two ~160 MB
    memsets per training step."""
    key = (int(M), int(width), bool(ones), bool(f16))
    nbytes = _PLANES_BYTES.get(key)
    if nbytes is None:
        nb = C.c_size_t()
        fn = load().clica_mlp_planes16_bytes if f16 else load().clica_mlp_planes_bytes      # f16x2: two pieces per unit
        check(fn(int(M), int(width), 1 if ones else 0, C.byref(nb)), "clica_mlp_planes_bytes")
        nbytes = _PLANES_BYTES[key] = nb.value
    return (torch.zeros if zero else torch.empty)(nbytes, dtype=torch.uint8, device=device)


def mlp_planes_from_f32(x: torch.Tensor, ones: bool, out: Optional[torch.Tensor] = None, state=None, tensor=None) -> torch.Tensor:
    """Plane copy (operand format of `mlp_wgrad_split`) of a fp32 [M, width] tensor: bf16x3 (clica_mlp_planes_from_f32), or -- with a
    `Split16` state and `tensor` = (family, index) in it -- f16x2 on that tensor's scale (clica_mlp_planes16_from_f32)."""
    (x, ldx) = _mat("x", x)
    M, width = x.shape
    if out is None:
        out = mlp_planes_alloc(M, width, ones, x.device, zero=False, f16=state is not None)
    if state is None:
        check(load().clica_mlp_planes_from_f32(x.data_ptr(), ldx, M, width, 1 if ones else 0, out.data_ptr(), stream_ptr()), "clica_mlp_planes_from_f32")
    else:
        check(load().clica_mlp_planes16_from_f32(x.data_ptr(), ldx, M, width, 1 if ones else 0, out.data_ptr(), state.buf.data_ptr(),
                                                 int(tensor[0]), int(tensor[1]), stream_ptr()), "clica_mlp_planes16_from_f32")
    return out


def mlp_planes_to_f32(buf: torch.Tensor, rows: int, feats: int, ones: bool) -> torch.Tensor:
    """Decode a bf16 plane buffer (csrc/planes.h) back into the fp32 [rows, feats] matrix it holds (hi + mid + lo is exact in fp32).
    Inspection / tests only -- plain tensor ops, no kernel of ours."""
    units = (feats + (1 if ones else 0) + 31) // 32
    raw = buf.view(torch.int16)
    groups = raw.numel() // (units * 3 * 512)
    a = (raw.view(groups, units, 3, 4, 2, 4, 16).to(torch.int32) << 16).view(torch.float32)      # [g][u][plane][k/4][f/16][k%4][f%16]
    v = (a[:, :, 0] + a[:, :, 1]) + a[:, :, 2]
    return v.permute(0, 2, 4, 1, 3, 5).reshape(groups * 16, units * 32)[:rows, :feats].contiguous()


def mlp_planes16_to_f32(buf: torch.Tensor, rows: int, feats: int, ones: bool, scale: float) -> torch.Tensor:
    """Decode an f16x2 plane buffer (two fp16 pieces per unit, scaled by `scale`) into the fp32 [rows, feats] matrix it represents:
    (hi + lo) / scale.  Inspection / tests only."""
    units = (feats + (1 if ones else 0) + 31) // 32
    raw = buf.view(torch.float16)
    groups = raw.numel() // (units * 2 * 512)
    a = raw.view(groups, units, 2, 4, 2, 4, 16).to(torch.float32)       # [g][u][piece][k/4][f/16][k%4][f%16]
    v = (a[:, :, 0] + a[:, :, 1]) / float(scale)
    return v.permute(0, 2, 4, 1, 3, 5).reshape(groups * 16, units * 32)[:rows, :feats].contiguous()


def mlp_planes_from_f32_t(x: torch.Tensor, out: Optional[torch.Tensor] = None, state=None, tensor=None) -> torch.Tensor:
    """T-planes of a fp32 [M, width] tensor = planes of its transpose (rows = feature, features = batch row): the A operand of
    `linear_split_fwd` / `linear_split_dgrad` (clica_mlp_planes_from_f32_t; f16x2 with `state` / `tensor` as in mlp_planes_from_f32)."""
    (x, ldx) = _mat("x", x)
    M, width = x.shape
    if out is None:
        out = mlp_planes_alloc(width, M, False, x.device, f16=state is not None)
    if state is None:
        check(load().clica_mlp_planes_from_f32_t(x.data_ptr(), ldx, M, width, out.data_ptr(), stream_ptr()), "clica_mlp_planes_from_f32_t")
    else:
        check(load().clica_mlp_planes16_from_f32_t(x.data_ptr(), ldx, M, width, out.data_ptr(), state.buf.data_ptr(), int(tensor[0]), int(tensor[1]),
                                                   stream_ptr()), "clica_mlp_planes16_from_f32_t")
    return out


def _pp(t):
    return t.data_ptr() if t is not None else None


def linear_split_fwd(xT, wT, bias, M: int, N: int, K: int, leaky: bool, slope: float, yT=None, yN=None, yN_ones: bool = True, y=None,
                     state=None, layer=None):
    """One wide nn.Linear (+ LeakyReLU) forward in split-bf16 arithmetic from T-plane operands (clica_linear_split_fwd)."""
    ldy = 0
    if y is not None:
        require_cuda(y, "y")
        if y.dim() != 2 or y.stride(1) != 1:
            raise ValueError("y must be a 2-D tensor with contiguous rows")
        ldy = y.stride(0)
    args = (xT.data_ptr(), wT.data_ptr(), _pp(bias), int(M), int(N), int(K), 1 if leaky else 0, float(slope), _pp(yT), _pp(yN),
            1 if yN_ones else 0, _pp(y), ldy)
    if state is None:
        check(load().clica_linear_split_fwd(*args, stream_ptr()), "clica_linear_split_fwd")
    else:      # f16x2: operands are the state's tensors A[layer] / W[layer], the output is A[layer + 1]
        check(load().clica_linear_split_fwd16(*args, state.buf.data_ptr(), int(layer), stream_ptr()), "clica_linear_split_fwd16")


def linear_split_dgrad(dzT, wN, actT, slope: float, M: int, N: int, K: int, dxT=None, dxN=None, dx=None, state=None, layer=None):
    """dX = (dZ W) * LeakyReLU'(layer input) of one wide layer in split-bf16 arithmetic (clica_linear_split_dgrad)."""
    ldd = 0
    if dx is not None:
        require_cuda(dx, "dx")
        if dx.dim() != 2 or dx.stride(1) != 1:
            raise ValueError("dx must be a 2-D tensor with contiguous rows")
        ldd = dx.stride(0)
    args = (dzT.data_ptr(), wN.data_ptr(), _pp(actT), float(slope), int(M), int(N), int(K), _pp(dxT), _pp(dxN), _pp(dx), ldd)
    if state is None:
        check(load().clica_linear_split_dgrad(*args, stream_ptr()), "clica_linear_split_dgrad")
    else:      # f16x2: operands D[layer] / W[layer], the output is D[layer - 1]
        check(load().clica_linear_split_dgrad16(*args, state.buf.data_ptr(), int(layer), stream_ptr()), "clica_linear_split_dgrad16")


def mlp_wgrad_split_kind(N: int, K: int) -> int:
    """0: the layer's weight gradient runs on the bf16 matrix cores from plane copies; 1: fp32 tiny-dimension kernel."""
    k = C.c_int32()
    check(load().clica_mlp_wgrad_split_kind(int(N), int(K), C.byref(k)), "clica_mlp_wgrad_split_kind")
    return int(k.value)


def mlp_fwd_split(x: torch.Tensor, weights, biases, outs, packed_split: torch.Tensor, slope: float = 0.01, signmasks=None, mix=None,
                  planes=None, state=None):
    """`mlp_fwd` on the bf16 matrix cores with exact 3-way bf16 splits (clica_mlp_fwd_split); `weights` only give the shapes.
    `planes[l]` (from mlp_planes_alloc(M, width_l, True), or None) receives layer l's output as bf16 planes for
    `mlp_wgrad_split`; `outs[l]` may then be None (no fp32 copy of that hidden activation)."""
    (x, ldx) = _mat("x", x)
    L = len(weights)
    bs = [None if b is None else b.detach().contiguous() for b in biases]
    VP, I64, I32 = C.c_void_p * L, C.c_int64 * L, C.c_int32 * L
    gW, gslope, xout = mix if mix is not None else (None, 0.0, None)
    if gW is not None:
        gW = gW.detach().contiguous()
    args = (x.data_ptr(), ldx, x.shape[0], ptr(gW), 0 if gW is None else gW.shape[0], float(gslope),
            ptr(xout), 0 if xout is None else xout.stride(0), L,
            VP(*[None if b is None else b.data_ptr() for b in bs]),
            VP(*[ptr(o) for o in outs]), I64(*[0 if o is None else o.stride(0) for o in outs]),
            I32(*[w.shape[0] for w in weights]), I32(*[w.shape[1] for w in weights]),
            packed_split.data_ptr(), None if signmasks is None else VP(*[ptr(m) for m in signmasks]),
            None if planes is None else VP(*[ptr(q) for q in planes]), float(slope))
    if state is None:
        check(load().clica_mlp_fwd_split(*args, stream_ptr()), "clica_mlp_fwd_split")
    else:      # f16x2 arithmetic on `state`'s scales (ops.Split16)
        check(load().clica_mlp_fwd_split16(*args, state.buf.data_ptr(), stream_ptr()), "clica_mlp_fwd_split16")
    return outs[-1]


def mlp_chain_tail_supported(shapes) -> bool:
    """Can the backward chain leave the weight-gradient slabs of the first / last layer of an encoder with these (out, in) shapes
    (forward order) itself?  (clica_mlp_chain_tail_supported)"""
    L = len(shapes)
    I32 = C.c_int32 * L
    ok = C.c_int32()
    check(load().clica_mlp_chain_tail_supported(L, I32(*[s[0] for s in shapes]), I32(*[s[1] for s in shapes]), C.byref(ok)),
          "clica_mlp_chain_tail_supported")
    return bool(ok.value)


def mlp_dgrad_chain_split(dy: torch.Tensor, weights_chain, packed_split_t: torch.Tensor, outs, slope: float = 0.01, masks_chain=None,
                          planes=None, state=None, tail: Optional[dict] = None):
    """`mlp_dgrad_chain` on the bf16 matrix cores (clica_mlp_dgrad_split); sign bits from `mlp_fwd_split`.  `planes[j]`
    (mlp_planes_alloc(M, width, False) or None) receives link j's dZ as bf16 planes; `outs[j]` may then be None.
    `tail` = dict(a_last, x, shapes, ws): the launch also leaves the weight-gradient slabs of the encoder's n-wide first and last
    layer in `ws`, the workspace of the `mlp_wgrad_split(..., adam=..., tail_slabs=True)` call that follows
    (clica_mlp_dgrad_split_tail)."""
    (dy, lddy) = _mat("dy", dy)
    n = len(weights_chain)
    I32, I64, VP = C.c_int32 * n, C.c_int64 * n, C.c_void_p * n
    args = (dy.data_ptr(), lddy, dy.shape[0], n,
            I32(*[w.shape[1] for w in weights_chain]), I32(*[w.shape[0] for w in weights_chain]),
            packed_split_t.data_ptr(), None if masks_chain is None else VP(*[ptr(m) for m in masks_chain]),
            VP(*[ptr(o) for o in outs]), I64(*[0 if o is None else o.stride(0) for o in outs]),
            None if planes is None else VP(*[ptr(q) for q in planes]), float(slope))
    if tail is not None:
        (al, lda), (xx, ldx) = _mat("a_last", tail["a_last"]), _mat("x", tail["x"])
        shapes = [tuple(sh) for sh in tail["shapes"]]
        Le = len(shapes)
        N32, K32 = (C.c_int32 * Le)(*[sh[0] for sh in shapes]), (C.c_int32 * Le)(*[sh[1] for sh in shapes])
        parts = tail.get("dy_parts")          # _lib.DyParts filled by clica_lp_loss_bwd_sym_train_parts: the chain's prologue finishes dy
        desc = _lib.ChainTail(a_last=al.data_ptr(), lda=lda, x=xx.data_ptr(), ldx=ldx, n_layers=Le, N=N32, K=K32,
                              wgrad_workspace=tail["ws"].data_ptr(), wgrad_workspace_bytes=tail["ws"].numel(),
                              dy_parts=C.pointer(parts) if parts is not None else None)
        check(load().clica_mlp_dgrad_split_tail(*args, None if state is None else state.buf.data_ptr(), C.byref(desc), stream_ptr()),
              "clica_mlp_dgrad_split_tail")
    elif state is None:
        check(load().clica_mlp_dgrad_split(*args, stream_ptr()), "clica_mlp_dgrad_split")
    else:
        check(load().clica_mlp_dgrad_split16(*args, state.buf.data_ptr(), stream_ptr()), "clica_mlp_dgrad_split16")
    return outs


def mlp_wgrad_split_workspace(M: int, shapes, device) -> torch.Tensor:
    L = len(shapes)
    I32 = C.c_int32 * L
    nb = C.c_size_t()
    check(load().clica_mlp_wgrad_split_workspace_bytes(int(M), L, I32(*[s[0] for s in shapes]), I32(*[s[1] for s in shapes]), C.byref(nb)),
          "clica_mlp_wgrad_split_workspace_bytes")
    return torch.zeros(nb.value, dtype=torch.uint8, device=device)


def mlp_wgrad_split(M: int, dz_planes, x_planes, dzs, xs, dWs, dbs, ws: Optional[torch.Tensor] = None, accumulate: bool = False,
                    state=None, a_index=None, d_index=None, adam: Optional[dict] = None, tail_slabs: bool = False):
    """Every layer's dW / db in the split-bf16 arithmetic (clica_mlp_wgrad_split).  Per layer EITHER the two plane buffers
    (`dz_planes[l]`, `x_planes[l]`: layers with `mlp_wgrad_split_kind` 0) OR the fp32 operands (`dzs[l]`, `xs[l]`: kind 1).
    `adam` = dict(param, grad, exp_avg, exp_avg_sq, step_dev, lr, beta1, beta2, eps, grad_scale, t_offset, s16): the trailing
    reduction launch also applies the optimizer to the arenas the dW / db views live in (clica_mlp_wgrad_split_adam)."""
    L = len(dWs)
    VP, I64, I32 = C.c_void_p * L, C.c_int64 * L, C.c_int32 * L
    dzm = [None if t is None else _mat("dz", t) for t in dzs]
    xm = [None if t is None else _mat("x", t) for t in xs]
    shapes = [tuple(w.shape) for w in dWs]
    if ws is None:
        ws = mlp_wgrad_split_workspace(M, shapes, dWs[0].device)
    args = (int(M), L, VP(*[ptr(q) for q in dz_planes]), VP(*[ptr(q) for q in x_planes]),
            VP(*[None if m is None else m[0].data_ptr() for m in dzm]), I64(*[0 if m is None else m[1] for m in dzm]),
            VP(*[None if m is None else m[0].data_ptr() for m in xm]), I64(*[0 if m is None else m[1] for m in xm]),
            VP(*[w.data_ptr() for w in dWs]), I64(*[w.stride(0) for w in dWs]),
            VP(*[ptr(b) for b in dbs]), I32(*[s[0] for s in shapes]), I32(*[s[1] for s in shapes]), 1 if accumulate else 0)
    if tail_slabs and adam is None:
        raise ValueError("mlp_wgrad_split(tail_slabs=True) needs adam=...: the slabs the chain left are consumed by clica_mlp_wgrad_split_adam")
    if adam is not None:
        global PARAM_EPOCH
        PARAM_EPOCH += 1
        if accumulate:
            raise ValueError("mlp_wgrad_split(adam=...): accumulate is not supported")
        s16 = adam.get("s16")
        desc = _lib.AdamDesc(param=adam["param"].data_ptr(), grad=adam["grad"].data_ptr(), exp_avg=adam["exp_avg"].data_ptr(),
                             exp_avg_sq=adam["exp_avg_sq"].data_ptr(), count=adam["param"].numel(), lr=float(adam["lr"]),
                             beta1=float(adam["beta1"]), beta2=float(adam["beta2"]), eps=float(adam["eps"]),
                             grad_scale=float(adam.get("grad_scale", 1.0)), step_dev=adam["step_dev"].data_ptr(),
                             t_offset=int(adam.get("t_offset", 1)), split16_state=None if s16 is None else s16.buf.data_ptr(),
                             n_layers=0 if s16 is None else s16.n_layers)
        check(load().clica_mlp_wgrad_split_adam(*args[:-1], None if state is None else state.buf.data_ptr(),
                                                None if state is None else I32(*[int(i) for i in a_index]),
                                                None if state is None else I32(*[int(i) for i in d_index]), C.byref(desc),
                                                1 if tail_slabs else 0, ws.data_ptr(), ws.numel(), stream_ptr()), "clica_mlp_wgrad_split_adam")
    elif state is None:
        check(load().clica_mlp_wgrad_split(*args, ws.data_ptr(), ws.numel(), stream_ptr()), "clica_mlp_wgrad_split")
    else:      # f16x2 plane copies: per layer the positions of its operands' scales in the state (include/clica.h)
        check(load().clica_mlp_wgrad_split16(*args, state.buf.data_ptr(), I32(*[int(i) for i in a_index]), I32(*[int(i) for i in d_index]),
                                             ws.data_ptr(), ws.numel(), stream_ptr()), "clica_mlp_wgrad_split16")
    return ws


def mlp_signmask_alloc(M: int, n_layers: int, device, zero: bool = True) -> list:
    """Per-layer opaque sign-bit buffers for mlp_fwd(signmasks=...) / mlp_dgrad_chain(masks_chain=...).  The forward writes every
    word of a buffer it is given (all waves and lanes of every workgroup), so `zero=False` (no fill launches) is safe for a buffer
    that goes straight into a forward call."""
    nbytes = C.c_size_t()
    check(load().clica_mlp_signmask_bytes(int(M), C.byref(nbytes)), "clica_mlp_signmask_bytes")
    mk = torch.zeros if zero else torch.empty
    return [mk(nbytes.value // 8, dtype=torch.int64, device=device) for _ in range(n_layers)]


def mlp_fwd(x: torch.Tensor, weights, biases, outs, slope: float = 0.01, packed: Optional[torch.Tensor] = None,
            signmasks=None, mix=None):
    """Whole Linear(+LeakyReLU) stack in one launch (clica_mlp_fwd); `outs[l]` receives layer l's output
    (saved activations; the last one is the result).  Widths <= 512, <= 8 layers.  `packed` = the same
    weights from `mlp_pack_weights` (faster weight streaming).  `signmasks[l]` (from mlp_signmask_alloc, or
    None) receives the (out > 0) bits of layer l for the one-launch backward.
    `mix = (gW [L, n, n], slope, x_out)`: `x` holds the LATENTS and the mixing net g runs in the kernel's prologue
    (clica_mlp_fwd_mixed); x_out receives g(x)."""
    (x, ldx) = _mat("x", x)
    L = len(weights)
    ws = [_mat(f"weight[{l}]", w) for l, w in enumerate(weights)]
    bs = [None if b is None else b.detach().contiguous() for b in biases]
    for o in outs:
        require_cuda(o, "out")
    VP = C.c_void_p * L
    I64 = C.c_int64 * L
    I32 = C.c_int32 * L
    common = (L, VP(*[w.data_ptr() for w, _ in ws]), I64(*[ld for _, ld in ws]),
              VP(*[None if b is None else b.data_ptr() for b in bs]),
              VP(*[o.data_ptr() for o in outs]), I64(*[o.stride(0) for o in outs]),
              I32(*[w.shape[0] for w, _ in ws]), I32(*[w.shape[1] for w, _ in ws]),
              ptr(packed), None if signmasks is None else VP(*[ptr(m) for m in signmasks]),
              float(slope), stream_ptr())
    if mix is None:
        check(load().clica_mlp_fwd(x.data_ptr(), ldx, x.shape[0], *common), "clica_mlp_fwd")
    else:
        gW, gslope, xout = mix
        gW = gW.detach().contiguous()
        check(load().clica_mlp_fwd_mixed(x.data_ptr(), ldx, x.shape[0], gW.data_ptr(), gW.shape[0], float(gslope),
                                         xout.data_ptr(), xout.stride(0), *common), "clica_mlp_fwd_mixed")
    return outs[-1]


def mlp_dgrad_chain(dy: torch.Tensor, weights_chain, packed_t: torch.Tensor, acts_chain, outs, slope: float = 0.01,
                    masks_chain=None):
    """Backward data chain in one launch (clica_mlp_dgrad).  `weights_chain` = encoder weights in chain order
    (layer L-1 first ... layer 1), `packed_t` = mlp_pack_weights(weights_chain, transpose=True), `acts_chain[j]`
    = saved activation that fed layer j of the chain, `outs[j]` = dZ of the layer below (written).
    `masks_chain[j]` = that activation's sign bits from mlp_fwd (used instead of re-reading the activation)."""
    (dy, lddy) = _mat("dy", dy)
    n = len(weights_chain)
    I32, I64, VP = C.c_int32 * n, C.c_int64 * n, C.c_void_p * n
    acts = [None if a is None else _mat("act", a) for a in acts_chain]
    check(load().clica_mlp_dgrad(dy.data_ptr(), lddy, dy.shape[0], n,
                                 I32(*[w.shape[1] for w in weights_chain]), I32(*[w.shape[0] for w in weights_chain]),
                                 packed_t.data_ptr(),
                                 VP(*[None if a is None else a[0].data_ptr() for a in acts]), I64(*[0 if a is None else a[1] for a in acts]),
                                 None if masks_chain is None else VP(*[ptr(m) for m in masks_chain]),
                                 VP(*[o.data_ptr() for o in outs]), I64(*[o.stride(0) for o in outs]),
                                 float(slope), stream_ptr()), "clica_mlp_dgrad")
    return outs


def linear_plan(op: str, M: int, N: int, K: int):
    """(tile_m, tile_n, waves, splits) of the kernel instance a Linear launch of this shape uses."""
    v = [C.c_int32() for _ in range(4)]
    check(load().clica_linear_plan({"fwd": 0, "dgrad": 1, "wgrad": 2}[op], M, N, K, *[C.byref(x) for x in v]), "clica_linear_plan")
    return tuple(x.value for x in v)


# ------------------------------------------------------------------------------- heads
def rescale_fwd(x, r):
    (x, ldx) = _mat("x", x)
    M, n = x.shape
    y = torch.empty((M, n), dtype=torch.float32, device=x.device)
    inv = torch.empty((M,), dtype=torch.float32, device=x.device)
    check(load().clica_rescale_fwd(x.data_ptr(), ldx, r.data_ptr(), y.data_ptr(), n, inv.data_ptr(), M, n, stream_ptr()),
          "clica_rescale_fwd")
    return y, inv


def rescale_bwd(x, r, inv, dy, need_dx=True, need_dr=True):
    (x, ldx), (dy, lddy) = _mat("x", x), _mat("dy", dy)
    M, n = x.shape
    dx = torch.empty((M, n), dtype=torch.float32, device=x.device) if need_dx else None
    part = torch.empty(((M + 255) // 256,), dtype=torch.float32, device=x.device) if need_dr else None
    check(load().clica_rescale_bwd(x.data_ptr(), ldx, r.data_ptr(), inv.data_ptr(), dy.data_ptr(), lddy, ptr(dx), n,
                                   ptr(part), M, n, stream_ptr()), "clica_rescale_bwd")
    return dx, (part.sum().reshape(1) if need_dr else None)


def softclip_fwd(x, bound):
    (x, ldx) = _mat("x", x)
    M, n = x.shape
    y = torch.empty((M, n), dtype=torch.float32, device=x.device)
    check(load().clica_softclip_fwd(x.data_ptr(), ldx, bound.data_ptr(), y.data_ptr(), n, M, n, stream_ptr()),
          "clica_softclip_fwd")
    return y


def softclip_bwd(x, bound, dy, need_dx=True, need_db=True):
    (x, ldx), (dy, lddy) = _mat("x", x), _mat("dy", dy)
    M, n = x.shape
    dx = torch.empty((M, n), dtype=torch.float32, device=x.device) if need_dx else None
    part = torch.empty(((M + 255) // 256, n), dtype=torch.float32, device=x.device) if need_db else None
    check(load().clica_softclip_bwd(x.data_ptr(), ldx, bound.data_ptr(), dy.data_ptr(), lddy, ptr(dx), n, ptr(part),
                                    M, n, stream_ptr()), "clica_softclip_bwd")
    return dx, (part.sum(0) if need_db else None)


def leaky_relu_fwd(x, slope: float = 0.01):
    (x, ldx) = _mat("x", x)
    M, n = x.shape
    y = torch.empty((M, n), dtype=torch.float32, device=x.device)
    check(load().clica_leaky_relu_fwd(x.data_ptr(), ldx, y.data_ptr(), n, M, n, float(slope), stream_ptr()), "clica_leaky_relu_fwd")
    return y


def leaky_relu_bwd(yact, dy, slope: float = 0.01):
    (yact, ldy), (dy, lddy) = _mat("yact", yact), _mat("dy", dy)
    M, n = yact.shape
    dx = torch.empty((M, n), dtype=torch.float32, device=dy.device)
    check(load().clica_leaky_relu_bwd(yact.data_ptr(), ldy, dy.data_ptr(), lddy, dx.data_ptr(), n, M, n, float(slope), stream_ptr()),
          "clica_leaky_relu_bwd")
    return dx


# ------------------------------------------------------------------------------- mixing / adam / sampler
MIX_ACT = {"leaky_relu": 0, "relu": 0, "elu": 1, "smooth_leaky_relu": 2, "softplus": 3}


def mixing_fwd(z: torch.Tensor, weights: torch.Tensor, slope: float = 0.2, out: Optional[torch.Tensor] = None, act_kind: int = 0):
    """x = W_L phi(... phi(W_1 z)); `weights` is a contiguous [L, n, n] stack (nn.Linear layout).  `act_kind` / `slope`:
    0 LeakyReLU(slope) (slope 0 = ReLU), 1 ELU(alpha = slope), 2 SmoothLeakyReLU(alpha = slope), 3 Softplus(beta = slope)."""
    (z, ldz) = _mat("z", z)
    require_cuda(weights, "weights")
    M, n = z.shape
    if weights.dim() != 3 or weights.shape[1:] != (n, n) or not weights.is_contiguous():
        raise ValueError(f"weights must be contiguous [L,{n},{n}], got {tuple(weights.shape)}")
    x = out if out is not None else torch.empty((M, n), dtype=torch.float32, device=z.device)
    check(load().clica_mixing_fwd_act(z.data_ptr(), ldz, weights.data_ptr(), weights.shape[0], int(act_kind), float(slope), x.data_ptr(),
                                      x.stride(0), M, n, stream_ptr()), "clica_mixing_fwd_act")
    return x


PARAM_EPOCH = 0     # bumped by every raw-pointer parameter update (adam_step): caches derived from weights (fragment-order packs) key on it,
                    # because a kernel writing through data_ptr() does not advance torch's per-tensor version counters


def adam_step(param, grad, exp_avg, exp_avg_sq, step_dev, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0,
              ticket: Optional[torch.Tensor] = None, t_offset: int = 1, s16=None):
    """In-place Adam on flat fp32 arenas; `step_dev` int32[1] = updates already applied.  With `ticket` (device
    int32[1], zero) the same launch also advances `step_dev` by one (clica_adam_step_tick)."""
    global PARAM_EPOCH
    PARAM_EPOCH += 1
    for nm, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        require_cuda(t, nm)
        if not t.is_contiguous():
            raise ValueError(f"{nm} must be contiguous")
    if s16 is not None and ticket is not None:     # ... and so does the counter's tick (the drop-in optimizer's step)
        if t_offset != 1:
            raise ValueError("adam_step(s16=..., ticket=...): the fused tick advances the counter AFTER the update (t_offset = 1)")
        check(load().clica_adam_step_s16_tick(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                              param.numel(), float(lr), float(beta1), float(beta2), float(eps), float(grad_scale),
                                              step_dev.data_ptr(), ticket.data_ptr(), s16.buf.data_ptr(), s16.n_layers, stream_ptr()),
              "clica_adam_step_s16_tick")
        return True
    if s16 is not None and ticket is None:     # the f16x2 arithmetic's scale update rides in the optimizer launch (clica_adam_step_s16)
        check(load().clica_adam_step_s16(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                         param.numel(), float(lr), float(beta1), float(beta2), float(eps), float(grad_scale),
                                         step_dev.data_ptr(), int(t_offset), s16.buf.data_ptr(), s16.n_layers, stream_ptr()), "clica_adam_step_s16")
        return True
    if ticket is None and t_offset != 1:       # the counter was already advanced earlier in the step
        check(load().clica_adam_step_at(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                        param.numel(), float(lr), float(beta1), float(beta2), float(eps), float(grad_scale),
                                        step_dev.data_ptr(), int(t_offset), stream_ptr()), "clica_adam_step_at")
    elif ticket is None:
        check(load().clica_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                     param.numel(), float(lr), float(beta1), float(beta2), float(eps), float(grad_scale),
                                     step_dev.data_ptr(), stream_ptr()), "clica_adam_step")
    else:
        check(load().clica_adam_step_tick(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                          param.numel(), float(lr), float(beta1), float(beta2), float(eps), float(grad_scale),
                                          step_dev.data_ptr(), ticket.data_ptr(), stream_ptr()), "clica_adam_step_tick")


def publish_host(src: torch.Tensor, host_dst: torch.Tensor, seq_dev: torch.Tensor, host_seq: torch.Tensor):
    """src (device fp32, <= 64 values) -> host_dst (pinned fp32), then ++seq_dev (device int32) -> host_seq (pinned int32) behind a
    system-scope release: a host read from the middle of a captured step (clica_publish_host)."""
    assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and 1 <= src.numel() <= 64
    assert host_dst.is_pinned() and host_dst.dtype == torch.float32 and host_dst.numel() >= src.numel()
    assert seq_dev.is_cuda and seq_dev.dtype == torch.int32 and host_seq.is_pinned() and host_seq.dtype == torch.int32
    check(load().clica_publish_host(src.data_ptr(), src.numel(), host_dst.data_ptr(), seq_dev.data_ptr(), host_seq.data_ptr(), stream_ptr()),
          "clica_publish_host")


def stamp(slot: torch.Tensor, which: int):
    """Device-side begin (0) / end (1) time stamp into `slot` (int64 [1 + 2 * capacity], zeros): usable inside graph capture."""
    check(load().clica_stamp(slot.data_ptr(), int(which), (slot.numel() - 1) // 2, stream_ptr()), "clica_stamp")


def clock_probe(n: int, period_us: float, stream: "torch.cuda.Stream") -> torch.Tensor:
    """Start the one-wave shader-clock probe on `stream` (a SIDE stream): n samples, `period_us` apart; returns the device buffer
    int64 [n, 2] = (wall clock in 100 MHz ticks, core-clock counter), complete once `stream` is (clica_clock_probe)."""
    buf = torch.zeros((int(n), 2), dtype=torch.int64, device="cuda")
    check(load().clica_clock_probe(buf.data_ptr(), int(n), max(1, int(round(period_us * 100))), stream.cuda_stream), "clica_clock_probe")
    return buf


def clock_between(samples_cpu, t0: int, t1: int) -> Optional[float]:
    """Shader clock in GHz over the wall-clock window [t0, t1] (100 MHz ticks) from the probe's samples (numpy int64 [n, 2], sorted):
    cycles between the first and the last sample inside the window / their distance in time; None with fewer than two samples."""
    import numpy as np
    w = samples_cpu[:, 0]
    lo, hi = int(np.searchsorted(w, t0, "left")), int(np.searchsorted(w, t1, "right")) - 1
    if hi - lo < 1 or w[lo] == 0:
        return None
    return float(samples_cpu[hi, 1] - samples_cpu[lo, 1]) / (float(w[hi] - w[lo]) * 10.0)       # cycles per 10 ns tick -> GHz


def stamp_brackets(slot: torch.Tensor):
    """Completed (begin, end) pairs of a stamp slot as absolute wall-clock ticks (100 MHz)."""
    v = slot.cpu()
    n = min(int(v[0]), (v.numel() - 1) // 2)
    return [(int(v[1 + 2 * i]), int(v[2 + 2 * i])) for i in range(n)]


def stamp_intervals_us(slot: torch.Tensor):
    """Completed (begin, end) pairs of a stamp slot as microseconds (100 MHz counter)."""
    v = slot.cpu()
    n = min(int(v[0]), (v.numel() - 1) // 2)
    return [(int(v[2 + 2 * i]) - int(v[1 + 2 * i])) / 100.0 for i in range(n)]


def tick(counter: torch.Tensor):
    check(load().clica_tick(counter.data_ptr(), stream_ptr()), "clica_tick")


def _pair_descs(space, marginal, conditional, n, marginal_mean, m_scale, m_p, c_scale, c_p, box, seed, stream_id):
    mk = lambda dist, scale, p, sid: _lib.SamplerDesc(space=SPACE[space], dist=DIST[dist], n=n, box_min=float(box[0]), box_max=float(box[1]),
                                                      scale=float(scale), shape_p=float(p), seed=int(seed) & (2**64 - 1),
                                                      stream_id=int(sid) & 0xFFFFFFFF)
    dm, dc = mk(marginal, m_scale, m_p, stream_id), mk(conditional, c_scale, c_p, stream_id + 1)
    ldmm = 0
    if marginal_mean is not None:
        require_cuda(marginal_mean, "marginal_mean")
        marginal_mean = marginal_mean.reshape(1, -1) if marginal_mean.dim() == 1 else marginal_mean
        marginal_mean, ldmm = rowmajor(marginal_mean.detach())
        if marginal_mean.shape[0] == 1:
            ldmm = 0
    return dm, dc, marginal_mean, ldmm


def sample_pair(space: str, marginal: str, conditional: str, n: int, size: int, z: torch.Tensor, zt: torch.Tensor,
                marginal_mean: Optional[torch.Tensor] = None, m_scale: float = 1.0, m_p: float = 2.0,
                c_scale: float = 1.0, c_p: float = 2.0, box=(0.0, 1.0), seed: int = 0, stream_id: int = 0,
                step_dev: Optional[torch.Tensor] = None):
    """z ~ marginal, zt ~ conditional(. | z) with Philox stream ids `stream_id` and `stream_id + 1` (clica_sample_pair:
    one launch for the coordinate-wise kinds; same numbers as two `sample` calls)."""
    dm, dc, marginal_mean, ldmm = _pair_descs(space, marginal, conditional, n, marginal_mean, m_scale, m_p, c_scale, c_p, box, seed, stream_id)
    check(load().clica_sample_pair(C.byref(dm), C.byref(dc), ptr(marginal_mean), ldmm, z.data_ptr(), z.stride(0), zt.data_ptr(),
                                   zt.stride(0), size, ptr(step_dev), stream_ptr()), "clica_sample_pair")
    return z, zt


def mlp_pack_split16_sample(weights, packed: torch.Tensor, packed_t: torch.Tensor, state, space: str, marginal: str, conditional: str, n: int,
                            size: int, z: torch.Tensor, zt: torch.Tensor, marginal_mean: Optional[torch.Tensor] = None, m_scale: float = 1.0,
                            m_p: float = 2.0, c_scale: float = 1.0, c_p: float = 2.0, box=(0.0, 1.0), seed: int = 0, stream_id: int = 0,
                            step_dev: Optional[torch.Tensor] = None):
    """`mlp_pack_split_both(weights, packed, packed_t, state)` and `sample_pair(...)` in ONE launch (clica_mlp_pack_split16_both_sample):
    the training step's two independent front launches; same results bit for bit.  `packed` / `packed_t` must exist."""
    L = len(weights)
    ws = [_mat(f"weight[{l}]", w) for l, w in enumerate(weights)]
    I32 = C.c_int32 * L
    Ns, Ks = I32(*[w.shape[0] for w, _ in ws]), I32(*[w.shape[1] for w, _ in ws])
    Wp, ldp = (C.c_void_p * L)(*[w.data_ptr() for w, _ in ws]), (C.c_int64 * L)(*[ld for _, ld in ws])
    dm, dc, marginal_mean, ldmm = _pair_descs(space, marginal, conditional, n, marginal_mean, m_scale, m_p, c_scale, c_p, box, seed, stream_id)
    check(load().clica_mlp_pack_split16_both_sample(L, Wp, ldp, Ns, Ks, packed.data_ptr(), packed_t.data_ptr(), state.buf.data_ptr(),
                                                    C.byref(dm), C.byref(dc), ptr(marginal_mean), ldmm, z.data_ptr(), z.stride(0), zt.data_ptr(),
                                                    zt.stride(0), size, ptr(step_dev), stream_ptr()), "clica_mlp_pack_split16_both_sample")
    return packed, packed_t


# ------------------------------------------------------------------------------- nearest neighbours
def nn_search(table: torch.Tensor, query: torch.Tensor, k: int = 1, want_dist: bool = True):
    """Exact k nearest table rows of every query row in squared L2 (clica_nn_search; faiss.IndexFlatL2.search semantics:
    ascending distances, int64 labels).  Returns (dist (Q, k) float32 | None, idx (Q, k) int64)."""
    (tab, ldt), (qry, ldq) = _mat("table", table), _mat("query", query)
    N, n = tab.shape
    Q = qry.shape[0]
    if qry.shape[1] != n:
        raise ValueError(f"query {tuple(qry.shape)} does not match the table {tuple(tab.shape)}")
    nbytes = C.c_size_t()
    check(load().clica_nn_search_workspace_bytes(Q, N, n, int(k), C.byref(nbytes)), "clica_nn_search_workspace_bytes")
    ws = workspace("nn_search", nbytes.value, tab.device)
    idx = torch.empty((Q, k), dtype=torch.int64, device=tab.device)
    dist = torch.empty((Q, k), dtype=torch.float32, device=tab.device) if want_dist else None
    check(load().clica_nn_search(tab.data_ptr(), ldt, N, qry.data_ptr(), ldq, Q, n, int(k), idx.data_ptr(), ptr(dist),
                                 ws.data_ptr(), ws.numel(), stream_ptr()), "clica_nn_search")
    return dist, idx


def kitti_gather_pairs(frames: torch.Tensor, first: torch.Tensor, second: torch.Tensor, latents: Optional[torch.Tensor] = None,
                       image_shape=None):
    """Interleaved float32 image batch [2 B, 1, H, W] (and labels [2 B, n_lat]) of the frame pairs (first[i], second[i]) from a uint8 frame
    table [F, H, W] in HBM (clica_kitti_gather_pairs: kitti_masks/dataset.py:90-142 on the device)."""
    if not frames.is_cuda:
        raise ClicaError(f"frames must live on the GPU (got device {frames.device}); cl_ica_amd runs only through its HIP kernels")
    if frames.dtype != torch.uint8 or not frames.is_contiguous():
        raise ValueError("frames must be a contiguous uint8 tensor [F, ...]")
    F = frames.shape[0]
    elems = frames[0].numel()
    first = first.to(device=frames.device, dtype=torch.int64).contiguous()
    second = second.to(device=frames.device, dtype=torch.int64).contiguous()
    B = first.numel()
    if second.numel() != B or B == 0:
        raise ValueError("first / second must hold the same, non-zero number of frame indices")
    shape = tuple(image_shape) if image_shape is not None else (1,) + tuple(frames.shape[1:])
    images = torch.empty((2 * B,) + shape, dtype=torch.float32, device=frames.device)
    labels = lat = None
    n_lat = 0
    if latents is not None:
        require_cuda(latents, "latents")
        lat = latents.detach().to(torch.float32).contiguous()
        n_lat = lat.shape[1]
        labels = torch.empty((2 * B, n_lat), dtype=torch.float32, device=frames.device)
    check(load().clica_kitti_gather_pairs(frames.data_ptr(), elems, F, first.data_ptr(), second.data_ptr(), B, images.data_ptr(),
                                          ptr(lat), n_lat, ptr(labels), stream_ptr()), "clica_kitti_gather_pairs")
    return images, labels


def moments(a: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """G = [a | b | 1]^T [a | b | 1] in fp64 (clica_moments): the (da + db + 1)^2 second-moment matrix every disentanglement score is a
    function of (cl_ica_amd/disentanglement_utils.py).  a [M, da], b [M, db] fp32 on the device; returns a device fp64 tensor."""
    (a, lda) = _mat("a", a)
    M, da = a.shape
    db, ldb, bp = 0, 0, None
    if b is not None:
        (b, ldb) = _mat("b", b)
        if b.shape[0] != M:
            raise ValueError(f"row counts differ: {tuple(a.shape)} vs {tuple(b.shape)}")
        db, bp = b.shape[1], b.data_ptr()
    d = da + db + 1
    if d > 129:
        # wider than one launch of the kernel covers (2 n + 1 <= 129, i.e. n <= 64): the same kernel over 64-column blocks of
        # X = [a | b], G[I, J] = X_I^T X_J from the cross block of the launch on (X_I, X_J), sums from its last column
        X = [(a, c0, min(c0 + 64, da)) for c0 in range(0, da, 64)] + ([(b, c0, min(c0 + 64, db)) for c0 in range(0, db, 64)] if b is not None else [])
        offs, o = [], 0
        for (_, c0, c1) in X:
            offs.append(o); o += c1 - c0
        G = torch.empty((d, d), dtype=torch.float64, device=a.device)
        G[-1, -1] = float(M)
        for i, (ti, i0, i1) in enumerate(X):
            for j in range(i, len(X)):
                tj, j0, j1 = X[j]
                blk = moments(ti[:, i0:i1], tj[:, j0:j1])
                ni, nj = i1 - i0, j1 - j0
                G[offs[i]:offs[i] + ni, offs[j]:offs[j] + nj] = blk[:ni, ni:ni + nj]
                G[offs[j]:offs[j] + nj, offs[i]:offs[i] + ni] = blk[:ni, ni:ni + nj].t()
                if j == i:
                    G[offs[i]:offs[i] + ni, offs[i]:offs[i] + ni] = blk[:ni, :ni]
                    G[offs[i]:offs[i] + ni, -1] = blk[:ni, -1]
                    G[-1, offs[i]:offs[i] + ni] = blk[-1, :ni]
        return G
    nbytes = C.c_size_t()
    check(load().clica_moments_workspace_bytes(M, d, C.byref(nbytes)), "clica_moments_workspace_bytes")
    ws = workspace("moments", nbytes.value, a.device)
    out = torch.empty((d, d), dtype=torch.float64, device=a.device)
    check(load().clica_moments(a.data_ptr(), lda, da, bp, ldb, db, M, out.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr()), "clica_moments")
    return out


SPACE = {"real": 0, "box": 1, "sphere": 2}
DIST = {"uniform": 0, "normal": 1, "laplace": 2, "gennorm": 3, "vmf": 4}


def sample(space: str, dist: str, n: int, size: int, device, mean: Optional[torch.Tensor] = None,
           scale: float = 1.0, shape_p: float = 2.0, box=(0.0, 1.0), seed: int = 0, stream_id: int = 0,
           step_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           scale_vec: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`scale_vec`: optional (n,), (1, n) or (size, n) tensor of per-coordinate scales multiplying `scale` (clica_sample_scaled)."""
    d = _lib.SamplerDesc(space=SPACE[space], dist=DIST[dist], n=n, box_min=float(box[0]), box_max=float(box[1]),
                         scale=float(scale), shape_p=float(shape_p), seed=int(seed) & (2**64 - 1),
                         stream_id=int(stream_id) & 0xFFFFFFFF)
    ldm = 0
    if mean is not None:
        require_cuda(mean, "mean")
        if mean.dim() == 1:
            mean = mean.reshape(1, -1)
        mean, ldm = rowmajor(mean.detach())
        if mean.shape[0] == 1:
            ldm = 0
        elif mean.shape[0] != size:
            raise ValueError(f"mean has {mean.shape[0]} rows, expected 1 or {size}")
    o = out if out is not None else torch.empty((size, n), dtype=torch.float32, device=device)
    if scale_vec is not None:
        require_cuda(scale_vec, "scale_vec")
        sv = scale_vec.detach().to(torch.float32)
        sv = sv.reshape(1, -1) if sv.dim() == 1 else sv
        if sv.dim() != 2 or sv.shape[1] != n or sv.shape[0] not in (1, size):
            raise ValueError(f"scale tensor of shape {tuple(scale_vec.shape)}: expected ({n},), (1, {n}) or ({size}, {n})")
        sv, lds = rowmajor(sv)
        check(load().clica_sample_scaled(C.byref(d), ptr(mean), ldm, sv.data_ptr(), 0 if sv.shape[0] == 1 else lds, o.data_ptr(),
                                         o.stride(0), size, ptr(step_dev), stream_ptr()), "clica_sample_scaled")
        return o
    check(load().clica_sample(C.byref(d), ptr(mean), ldm, o.data_ptr(), o.stride(0), size, ptr(step_dev), stream_ptr()),
          "clica_sample")
    return o
