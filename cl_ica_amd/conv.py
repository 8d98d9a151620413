"""Conv stack of the KITTI-masks encoder on the HIP library (``clica_conv_*``, csrc/linear.hip, conv section).

Replaces the five ``nn.Conv2d(k = 4) + ReLU`` stages of ``BetaVAE_H`` (/root/reference/kitti_masks/model.py:41-56) -- forward,
data gradients, weight and bias gradients -- for (images, nc, 64, 64) inputs: four implicit-GEMM stride-2 stages that hand
their output to the next stage as its padded space-to-depth tensor (channels-last, no NCHW <-> NHWC copies, bias + ReLU in the
GEMM epilogue, ReLU gate in the data-gradient epilogue) and the k = 4 stage on the 4 x 4 map as ``clica_linear_*`` over the
flattened map.  The ``nn.Conv2d`` modules stay the owners of the parameters (state dict = the reference's); per call the
weights are re-ordered into the GEMM layouts (a few tiny copies) and the gradients come back in ``Conv2d.weight`` layout.

Arithmetic of the three 16C-deep stages (forward, data and weight gradients): ``f16x2`` (default; csrc/conv16.hip -- three fp16 matrix
products of two-piece operands per fp32 product, per-tensor power-of-two scales measured in the same step, fp32 accumulation) or
``f32`` (``CLICA_CONV_ARITH=f32`` / ``set_arith("f32")``: the fp32-MFMA kernels of csrc/linear.hip).  The first stage (K = 16) and the
4 x 4 stage stay fp32 kernels in both.

``conv_stack(x, w1, b1, ..., w5, b5) -> (images, 256)`` is one autograd node.  The activations it saves live in a per-shape
buffer set that is zeroed once (borders / non-output rows are never written) and handed back by the backward pass.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Tuple

import torch

from . import ops
from .encoders import _inplace_ok
from ._lib import check, load, ptr, stream_ptr, workspace

__all__ = ["conv_stack", "STAGES", "set_arith", "get_arith"]

_ARITH = os.environ.get("CLICA_CONV_ARITH", "f16x2")
if _ARITH not in ("f16x2", "f32"):
    raise ValueError(f"CLICA_CONV_ARITH={_ARITH!r}: 'f16x2' or 'f32'")
_SLOTS = 256          # csrc/conv16.hip: kSlots
_FIRST_FROM_IMAGE = True     # the one-channel first stage reads the images themselves (test hook: False = from the patch matrix, tests/test_gpu_conv.py)
_FIRST_MFMA = True           # f16x2: its forward on the matrix cores too


def set_arith(name: str) -> str:
    """Select the arithmetic of the 16C-deep stages ('f16x2' or 'f32'); returns the previous one."""
    global _ARITH
    if name not in ("f16x2", "f32"):
        raise ValueError(f"conv arithmetic {name!r}: 'f16x2' or 'f32'")
    prev, _ARITH = _ARITH, name
    return prev


def get_arith() -> str:
    return _ARITH

# (out_channels, spatial size of the output) of the four stride-2 stages for a 64 x 64 input; the fifth stage maps the
# 4 x 4 x 64 map to 256 features
STAGES = ((32, 32), (32, 16), (64, 8), (64, 4))
_FEATURES = 256
_IMAGE = 64


class _Buffers:
    """Activations / gradients of one in-flight forward-backward pair for a given (images, nc, device)."""

    def __init__(self, images: int, nc: int, device):
        f32 = dict(dtype=torch.float32, device=device)
        self.images, self.nc = images, nc
        self._patches = None                            # patch matrix of the first stage, allocated on first use (nc = 1 reads the images instead)
        self.S: Dict[int, torch.Tensor] = {}        # stage index (1..3) -> its space-to-depth INPUT, flat, zero tail
        self.dO: Dict[int, torch.Tensor] = {}       # stage index (0..3) -> gradient of its pre-activation output on its row grid
        self._dO_store: Dict[int, torch.Tensor] = {}
        self.gate: Dict[int, torch.Tensor] = {}     # stage index (0..2) -> (output > 0) bits on the stage's row grid, [row][cout / 32] words
        cin = nc
        for l, (cout, ho) in enumerate(STAGES):
            if l < 3:
                grid = ho if l == 0 else ho + 1
                self.gate[l] = torch.empty(images * grid * grid * (cout // 32), dtype=torch.int32, device=device)
            if l >= 1:
                hs = STAGES[l - 1][1] // 2 + 1
                self.S[l] = torch.zeros(images * hs * hs * 4 * cin + (hs + 2) * 4 * cin, **f32)
                front = (hs + 1) * cout
                store = torch.zeros(front + images * hs * hs * cout, **f32)
                self._dO_store[l] = store
                self.dO[l] = store[front:]
            else:
                self.dO[0] = torch.empty(images * ho * ho * cout, **f32)
            cin = cout
        self.wpack = None                               # GEMM-layout weights of the step in flight (conv._maps order)
        # f16x2 arithmetic: maxima slots of S1, S2, S3, dO3, dO2, dO1 (float bits), packed weight pieces + scales of W2g, W3g, W4g, W2dT, W3dT, W4dT
        self.amax = torch.zeros(7 * _SLOTS, dtype=torch.int32, device=device)     # (slot array 6: the input images, first stage on the matrix cores)
        self.w16 = None
        self.wscale = torch.ones(7, **f32)
        self.O4 = torch.zeros((images, 5 * 5 * 64), **f32)     # last stride-2 stage's output on its 5 x 5 row grid (non-output rows stay 0)
        # first stage's weight gradient: the small-matrix streaming kernel where its shape fits (nc = 1), the grouped GEMM path otherwise
        self.ws1 = self.ws1p = None
        if 256 % ((STAGES[0][0] // 4) * (16 * nc // 4)) == 0:
            nbytes = C.c_size_t()
            check(load().clica_conv_k4s2_wgrad_patches_workspace_bytes(images * 32 * 32, STAGES[0][0], 16 * nc, C.byref(nbytes)),
                  "clica_conv_k4s2_wgrad_patches_workspace_bytes")
            self.ws1p = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        else:
            self.ws1 = ops.mlp_wgrad_workspace(images * 32 * 32, [(STAGES[0][0], 16 * nc)], device)


    def patches_(self) -> torch.Tensor:
        if self._patches is None:
            self._patches = torch.empty((self.images * 32 * 32, 16 * self.nc), dtype=torch.float32, device=self.amax.device)
        return self._patches


_POOL: Dict[Tuple, List[_Buffers]] = {}


def _take(images: int, nc: int, device) -> _Buffers:
    free = _POOL.setdefault((images, nc, device.index), [])
    return free.pop() if free else _Buffers(images, nc, device)


def _give(buf: _Buffers, device) -> None:
    free = _POOL.setdefault((buf.images, buf.nc, device.index), [])
    if len(free) < 2:
        free.append(buf)


def _wg(w: torch.Tensor) -> torch.Tensor:
    """Conv2d weight [co][c][ky][kx] -> GEMM rows [co][(dy, dx, py, px, c)], ky = 2 dy + py, kx = 2 dx + px."""
    co, c = w.shape[:2]
    return w.detach().view(co, c, 2, 2, 2, 2).permute(0, 2, 4, 3, 5, 1).reshape(co, 16 * c)


def _wd(w: torch.Tensor) -> torch.Tensor:
    """Conv2d weight -> data-gradient operand [(1 - dy, 1 - dx, co)][(py, px, c)]."""
    co, c = w.shape[:2]
    return w.detach().view(co, c, 2, 2, 2, 2).flip(2, 4).permute(2, 4, 0, 3, 5, 1).reshape(4 * co, 4 * c)


def _wg_to_conv(dwg: torch.Tensor, co: int, c: int) -> torch.Tensor:
    return dwg.view(co, 2, 2, 2, 2, c).permute(0, 5, 1, 3, 2, 4).reshape(co, c, 4, 4)


def _w5(w: torch.Tensor) -> torch.Tensor:
    """[256][64][4][4] -> [256][(y, x, c) over the 5 x 5 row grid of the stage in front], zero columns for the non-output rows."""
    g = w.detach().permute(0, 2, 3, 1)
    return torch.nn.functional.pad(g, (0, 0, 0, 1, 0, 1)).reshape(w.shape[0], 5 * 5 * w.shape[1])


def _w5_back(dw5g: torch.Tensor) -> torch.Tensor:
    return dw5g.view(dw5g.shape[0], 5, 5, -1)[:, :4, :4, :].permute(0, 3, 1, 2)


_MAPS: Dict[Tuple, dict] = {}


def _maps(nc: int, device) -> dict:
    """Index maps of every weight re-ordering of a step, built once by running the layout functions above on index tensors:
    ``pack`` -- Conv2d.weight -> (W1g, W2g, W3g, W4g, W2d, W3d, W4d, W5g), ``unpack`` -- (dW1g .. dW5g) -> Conv2d.weight layout."""
    key = (nc, device.index)
    m = _MAPS.get(key)
    if m is not None:
        return m
    shapes = [(STAGES[0][0], nc, 4, 4), (32, 32, 4, 4), (64, 32, 4, 4), (64, 64, 4, 4), (_FEATURES, 64, 4, 4)]

    def idx(shape):
        n = 1
        for d in shape:
            n *= d
        return torch.arange(n, dtype=torch.int64).view(shape)

    def i32(t):
        return t.reshape(-1).to(torch.int32).to(device)

    w = [idx(sh) for sh in shapes]
    pack = [w[0].permute(0, 2, 3, 1).reshape(shapes[0][0], 16 * nc)] + [_wg(w[l]) for l in (1, 2, 3)] + [_wd(w[l]) for l in (1, 2, 3)]
    pack.append(torch.nn.functional.pad(w[4].permute(0, 2, 3, 1), (0, 0, 0, 1, 0, 1), value=-1).reshape(_FEATURES, 5 * 5 * 64))
    pack_src = [0, 1, 2, 3, 1, 2, 3, 4]
    gshapes = [(shapes[0][0], 16 * nc), (32, 512), (64, 512), (64, 1024), (_FEATURES, 5 * 5 * 64)]
    g = [idx(sh) for sh in gshapes]
    unpack = [g[0].view(shapes[0][0], 4, 4, nc).permute(0, 3, 1, 2)] + [_wg_to_conv(g[l], shapes[l][0], shapes[l][1]) for l in (1, 2, 3)] + [_w5_back(g[4])]
    pack16 = [pack[l] for l in (1, 2, 3)] + [pack[l].t().contiguous() for l in (4, 5, 6)] + [pack[0]]     # Wg as they are, Wd TRANSPOSED ([4 C][4 Cout]), W1g
    m = {"pack": [i32(t) for t in pack], "pack_shapes": [tuple(t.shape) for t in pack], "pack_src": pack_src,
         "unpack": [i32(t) for t in unpack], "shapes": shapes,
         "pack16": [i32(t) for t in pack16], "pack16_shapes": [tuple(t.shape) for t in pack16], "pack16_src": [1, 2, 3, 1, 2, 3, 0]}
    _MAPS[key] = m
    return m


def _gather(srcs, maps, dsts, accumulate=False):
    n = len(srcs)
    VP, I32 = C.c_void_p * n, C.c_int32 * n
    check(load().clica_conv_gather(n, VP(*[t.data_ptr() for t in srcs]), VP(*[None if t is None else t.data_ptr() for t in maps]),
                                   VP(*[t.data_ptr() for t in dsts]), I32(*[t.numel() for t in dsts]), int(accumulate), stream_ptr()),
          "clica_conv_gather")


def _pack16(buf: "_Buffers", ws_, m: dict) -> None:
    """Conv2d weights -> packed f16 hi / lo planes of the six GEMM operands + their scales (one launch)."""
    dev = buf.amax.device
    if buf.w16 is None:
        buf.w16 = [torch.empty(2 * sh[0] * sh[1], dtype=torch.int16, device=dev) for sh in m["pack16_shapes"]]
    n = len(m["pack16_src"])
    srcs = [ws_[i].detach() for i in m["pack16_src"]]
    srcs = [t if t.is_contiguous() else t.contiguous() for t in srcs]
    VP, I32 = C.c_void_p * n, C.c_int32 * n
    check(load().clica_conv16_pack(n, VP(*[t.data_ptr() for t in srcs]), I32(*[t.numel() for t in srcs]), VP(*[t.data_ptr() for t in m["pack16"]]),
                                   VP(*[t.data_ptr() for t in buf.w16]), I32(*[t.numel() for t in m["pack16"]]), buf.wscale.data_ptr(), stream_ptr()),
          "clica_conv16_pack")


def _slots(buf: "_Buffers", i: int) -> int:
    return buf.amax.data_ptr() + 4 * _SLOTS * i


class _ConvStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep, *params):
        lib, st = load(), stream_ptr()
        ws_, bs_ = params[0::2], params[1::2]
        images, nc = x.shape[0], x.shape[1]
        dev = x.device
        buf = _take(images, nc, dev)
        x = x.detach().contiguous()
        from_image = nc == 1 and _FIRST_FROM_IMAGE      # one input channel: the first stage reads the images themselves, no patch matrix
        if not from_image:
            check(lib.clica_conv_im2col_k4s2(x.data_ptr(), images, nc, _IMAGE, _IMAGE, buf.patches_().data_ptr(), st), "clica_conv_im2col_k4s2")
        m = _maps(nc, dev)
        if buf.wpack is None:
            buf.wpack = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in m["pack_shapes"]]
        srcs = [ws_[i].detach() for i in m["pack_src"]]
        if not all(t.is_contiguous() for t in srcs):
            srcs = [t.contiguous() for t in srcs]
        f16 = _ARITH == "f16x2"
        if f16:
            _gather([srcs[0], srcs[7]], [m["pack"][0], m["pack"][7]], [buf.wpack[0], buf.wpack[7]])     # the two fp32 stages' weights
            check(lib.clica_conv16_zero_slots(buf.amax.data_ptr(), 7, st), "clica_conv16_zero_slots")
            _pack16(buf, ws_, m)
        else:
            _gather(srcs, m["pack"], buf.wpack)          # W1g, W2g, W3g, W4g, W2d, W3d, W4d, W5g in one launch
        w1g = buf.wpack[0]
        in_kernel = f16 and nc == 1        # the K = 16 masks kernel records its output's maximum itself; other first stages get a pass of their own
        if from_image and f16 and _FIRST_MFMA:
            # the first stage too as three fp16 matrix products (csrc/conv16.hip: fwd_first16_k); the images' own maximum gives their scale
            if x.data_ptr() % 16:
                x = x.clone()
            check(lib.clica_conv16_amax(x.data_ptr(), x.numel(), _slots(buf, 6), st), "clica_conv16_amax")
            check(lib.clica_conv16_first_fwd(x.data_ptr(), buf.w16[6].data_ptr(), buf.wscale.data_ptr() + 4 * 6, ptr(bs_[0].detach()), images, _IMAGE, _IMAGE,
                                             STAGES[0][0], 1, buf.S[1].data_ptr(), buf.gate[0].data_ptr(), _slots(buf, 6), _slots(buf, 0), st),
                  "clica_conv16_first_fwd")
        elif from_image:
            check(lib.clica_conv_k4s2_fwd_image(x.data_ptr(), w1g.data_ptr(), ptr(bs_[0].detach()), images, _IMAGE, _IMAGE, STAGES[0][0], 1,
                                                buf.S[1].data_ptr(), buf.gate[0].data_ptr(), _slots(buf, 0) if in_kernel else None, st),
                  "clica_conv_k4s2_fwd_image")
        else:
            check(lib.clica_conv_k4s2_fwd_patches_amax(buf.patches_().data_ptr(), w1g.data_ptr(), ptr(bs_[0].detach()), images, 16 * nc, STAGES[0][0],
                                                       32, 32, 1, 1, buf.S[1].data_ptr(), buf.gate[0].data_ptr(), _slots(buf, 0) if in_kernel else None, st),
                  "clica_conv_k4s2_fwd_patches")
        if f16 and not in_kernel:
            check(lib.clica_conv16_amax(buf.S[1].data_ptr(), buf.S[1].numel(), _slots(buf, 0), st), "clica_conv16_amax")
        cin = STAGES[0][0]
        for l in (1, 2, 3):
            cout, ho = STAGES[l]
            hs = ho + 1
            out = buf.S[l + 1] if l < 3 else buf.O4
            if f16:
                check(lib.clica_conv16_k4s2_fwd(buf.S[l].data_ptr(), buf.w16[l - 1].data_ptr(), buf.wscale.data_ptr() + 4 * (l - 1), ptr(bs_[l].detach()),
                                                images, cin, cout, hs, hs, 1, 1 if l < 3 else 2, out.data_ptr(),
                                                buf.gate[l].data_ptr() if l < 3 else None, _slots(buf, l - 1), _slots(buf, l) if l < 3 else None, st),
                      "clica_conv16_k4s2_fwd")
            else:
                check(lib.clica_conv_k4s2_fwd(buf.S[l].data_ptr(), buf.wpack[l].data_ptr(), ptr(bs_[l].detach()), images, cin, cout, hs, hs,
                                              1, 1 if l < 3 else 2, out.data_ptr(), buf.gate[l].data_ptr() if l < 3 else None, st), "clica_conv_k4s2_fwd")
            cin = cout
        w5g = buf.wpack[7]
        feats = torch.empty((images, _FEATURES), dtype=torch.float32, device=dev)
        check(lib.clica_conv_k4s2_fwd_patches(buf.O4.data_ptr(), w5g.data_ptr(), ptr(bs_[4].detach()), images, 5 * 5 * 64, _FEATURES, 1, 1, 1, 0,
                                              feats.data_ptr(), None, st), "clica_conv_k4s2_fwd_patches")
        if keep:
            ctx.buf, ctx.w5g, ctx.nc, ctx.f16 = buf, w5g, nc, f16
            ctx.x_image = x if from_image else None
            ctx.params = params
            ctx.save_for_backward(feats, *ws_)
        else:
            _give(buf, dev)
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        lib, st = load(), stream_ptr()
        feats, *ws_ = ctx.saved_tensors
        buf, nc = ctx.buf, ctx.nc
        if buf is None:
            raise RuntimeError("conv_stack: backward called twice (the saved activations were handed back to the pool)")
        ctx.buf = None
        images, dev = buf.images, feats.device
        grads: List = [None] * 10
        dpre = ops.leaky_relu_bwd(feats, dfeats.contiguous(), slope=0.0)
        dw5g, db5 = ops.linear_wgrad(dpre, buf.O4)
        grads[9] = db5
        dwg_all = [None, None, None, None, dw5g]
        ops.linear_dgrad(dpre, ctx.w5g, buf.O4, slope=0.0, out=buf.dO[3].view(images, 5 * 5 * 64))
        f16 = ctx.f16
        if f16:
            check(lib.clica_conv16_amax(buf.dO[3].data_ptr(), images * 5 * 5 * 64, _slots(buf, 3), st), "clica_conv16_amax")
        for l in (3, 2, 1):
            cout, ho = STAGES[l]
            cin, hs = STAGES[l - 1][0], ho + 1
            nbytes = C.c_size_t()
            if f16:
                check(lib.clica_conv16_k4s2_wgrad_workspace_bytes(images * hs * hs, cout, 16 * cin, C.byref(nbytes)), "clica_conv16_k4s2_wgrad_workspace_bytes")
                wsp = workspace("conv16_wgrad", nbytes.value, dev)
                dwg = torch.empty((cout, 16 * cin), dtype=torch.float32, device=dev)
                db = torch.empty((cout,), dtype=torch.float32, device=dev)
                sd = 3 + (3 - l)                                 # slots of dO[l]
                check(lib.clica_conv16_k4s2_wgrad(buf.dO[l].data_ptr(), buf.S[l].data_ptr(), images, cin, cout, hs, hs, dwg.data_ptr(), db.data_ptr(),
                                                  0, _slots(buf, sd), _slots(buf, l - 1), wsp.data_ptr(), wsp.numel(), st), "clica_conv16_k4s2_wgrad")
                dwg_all[l], grads[2 * l + 1] = dwg, db
                dgrid = STAGES[l - 1][1] + (1 if l > 1 else 0)
                check(lib.clica_conv16_k4s2_dgrad(buf.dO[l].data_ptr(), buf.w16[3 + l - 1].data_ptr(), buf.wscale.data_ptr() + 4 * (3 + l - 1), images, cin, cout,
                                                  hs, hs, buf.dO[l - 1].data_ptr(), dgrid, dgrid, buf.gate[l - 1].data_ptr(),
                                                  _slots(buf, sd), _slots(buf, sd + 1) if l > 1 else None, st), "clica_conv16_k4s2_dgrad")
                continue
            check(lib.clica_conv_k4s2_wgrad_workspace_bytes(images * hs * hs, cout, 16 * cin, C.byref(nbytes)), "clica_conv_k4s2_wgrad_workspace_bytes")
            wsp = workspace("conv_wgrad", nbytes.value, dev)
            dwg = torch.empty((cout, 16 * cin), dtype=torch.float32, device=dev)
            db = torch.empty((cout,), dtype=torch.float32, device=dev)
            check(lib.clica_conv_k4s2_wgrad(buf.dO[l].data_ptr(), buf.S[l].data_ptr(), images, cin, cout, hs, hs, dwg.data_ptr(), db.data_ptr(),
                                            0, wsp.data_ptr(), wsp.numel(), st), "clica_conv_k4s2_wgrad")
            dwg_all[l], grads[2 * l + 1] = dwg, db
            dgrid = STAGES[l - 1][1] + (1 if l > 1 else 0)          # previous stage's row grid (the first stage's is its 32 x 32 output)
            check(lib.clica_conv_k4s2_dgrad(buf.dO[l].data_ptr(), buf.wpack[3 + l].data_ptr(), buf.S[l].data_ptr(), images, cin, cout, hs, hs,
                                            buf.dO[l - 1].data_ptr(), dgrid, dgrid, buf.gate[l - 1].data_ptr(), st), "clica_conv_k4s2_dgrad")
        cout = STAGES[0][0]
        dw1g = torch.empty((cout, 16 * nc), dtype=torch.float32, device=dev)
        db1 = torch.empty((cout,), dtype=torch.float32, device=dev)
        if ctx.x_image is not None:
            check(lib.clica_conv_k4s2_wgrad_image(buf.dO[0].data_ptr(), ctx.x_image.data_ptr(), images, _IMAGE, _IMAGE, cout, dw1g.data_ptr(),
                                                  db1.data_ptr(), 0, buf.ws1p.data_ptr(), buf.ws1p.numel(), st), "clica_conv_k4s2_wgrad_image")
        elif buf.ws1p is not None:
            check(lib.clica_conv_k4s2_wgrad_patches(buf.dO[0].data_ptr(), buf.patches_().data_ptr(), images * 32 * 32, cout, 16 * nc, dw1g.data_ptr(),
                                                    db1.data_ptr(), 0, buf.ws1p.data_ptr(), buf.ws1p.numel(), st), "clica_conv_k4s2_wgrad_patches")
        else:
            ops.mlp_wgrad([buf.dO[0].view(-1, cout)], [buf.patches_()], [dw1g], [db1], ws=buf.ws1)
        dwg_all[0], grads[1] = dw1g, db1
        m = _maps(nc, dev)
        prm = ctx.params
        if _inplace_ok(prm, (True, True) + tuple(ctx.needs_input_grad[2:])):
            # flat optimizer + a plain loss.backward(): the ten gradients are ADDED into its .grad views by one launch (what autograd's
            # AccumulateGrad nodes would do with ten adds) and autograd is handed None -- same rule as the MLP encoders (encoders._inplace_ok)
            _gather(dwg_all + [grads[2 * l + 1] for l in range(5)], m["unpack"] + [None] * 5,
                    [prm[2 * l].grad for l in range(5)] + [prm[2 * l + 1].grad for l in range(5)], accumulate=True)
            grads = [None] * 10
        else:
            out = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in m["shapes"]]
            _gather(dwg_all, m["unpack"], out)           # the five weight gradients back in Conv2d.weight layout, one launch
            for l in range(5):
                grads[2 * l] = out[l]
        dx = None
        if ctx.needs_input_grad[0]:      # d loss / d image (never in a training step: the encoder's input is data)
            dx = torch.empty((images, nc, _IMAGE, _IMAGE), dtype=torch.float32, device=dev)
            w1 = ws_[0].detach().contiguous()
            check(lib.clica_conv_k4s2_dgrad_input(buf.dO[0].data_ptr(), w1.data_ptr(), images, nc, STAGES[0][0], _IMAGE, _IMAGE, dx.data_ptr(), st),
                  "clica_conv_k4s2_dgrad_input")
        _give(buf, dev)
        return (dx, None, *grads)


def conv_stack(x: torch.Tensor, convs) -> torch.Tensor:
    """``convs`` = the five ``nn.Conv2d`` modules of ``BetaVAE_H.encoder``; ``x`` = (images, nc, 64, 64) float32 on the GPU."""
    if x.dim() != 4 or x.shape[2] != _IMAGE or x.shape[3] != _IMAGE:
        raise ValueError(f"conv_stack: input must be (images, nc, {_IMAGE}, {_IMAGE}), got {tuple(x.shape)}")
    if x.dtype != torch.float32:
        raise TypeError(f"conv_stack: float32 input expected, got {x.dtype}")
    if x.requires_grad and torch.is_grad_enabled() and x.shape[1] > 4:
        raise NotImplementedError("conv_stack: the input-image gradient kernel covers nc <= 4 channels")
    params = []
    for m in convs:
        if m.bias is None:
            raise ValueError("conv_stack: Conv2d stages without bias are not supported")
        params += [m.weight, m.bias]
    # `keep`: a backward pass may follow (forward() itself always runs with grad mode off, so it cannot tell); without one the
    # buffer set goes straight back to the pool
    keep = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    return _ConvStackFn.apply(x, keep, *params)
