#!/usr/bin/env python3
"""VERDICT r2 item 2 ("hide the loss behind MFMA work"), measured.  The only MFMA work of the step whose inputs exist before the
loss backward has finished is the z~ half of the backward data chain: the gradient w.r.t. z2_rec is the positive-pair term
alone and is written by the loss FORWARD's finalize, so rows [B, 2B) of the chain can start while the pair sweep of the loss
backward (VALU-only, ~55 us) is still producing the gradient of rows [0, B).  Variants, each captured into a HIP graph and
replayed:
   base   : the engine's step (chain = one launch of 256 workgroups behind the loss backward)
   split  : chain rows [B, 2B) on a side stream concurrently with the loss backward sweep, rows [0, B) behind the sweep
   serial : the same two half launches, but one after the other on one stream (cost of splitting alone)
usage: python tools/overlap_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl_ica_amd import _lib, encoders, ops
from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec

n, B = 10, 6144
torch.manual_seed(0)
f = encoders.get_mlp(n, n, [n * 10, n * 50, n * 50, n * 50, n * 50, n * 10]).to("cuda")
gW = torch.randn(3, n, n, device="cuda") / n ** 0.5
tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=1e-4, device="cuda", split_bf16=True)
assert tr.split_wgrad and tr.loss_train
L, R = len(tr.linears), 2 * B
chain = list(range(L - 1, 0, -1))
ws = [tr.linears[l].weight for l in chain]
groups_alloc = (R + 47) // 48 * 3


def rows(t, r0, r1):
    return None if t is None else t[r0:r1]


def plane_rows(buf, r0, r1):
    if buf is None:
        return None
    per_group = buf.numel() // groups_alloc
    return buf[(r0 // 16) * per_group:(r1 // 16) * per_group]


def mask_rows(m, r0, r1):
    return None if m is None else m[(r0 // 48) * 512:(r1 // 48) * 512]


def chain_rows(g, r0, r1):
    ops.mlp_dgrad_chain_split(g[r0:r1], ws, tr.packed_t, [rows(tr.dz_out[l - 1], r0, r1) for l in chain], tr.slope,
                              masks_chain=[mask_rows(tr.signmasks[l - 1], r0, r1) for l in chain],
                              planes=[plane_rows(tr.dz_planes[l - 1], r0, r1) for l in chain])


def loss_two_calls(between):
    """tr.loss_forward_backward with a hook between the forward (which writes dy[B:]) and the backward sweep"""
    lib, st = _lib.load(), _lib.stream_ptr()
    o = tr.loss_out
    y1, y2 = tr.y[:B], tr.y[B:]
    lse = o[2 * B:3 * B]
    _lib.check(lib.clica_lp_loss_fwd_train(C.byref(tr.desc), y1.data_ptr(), n, y2.data_ptr(), n, y1.data_ptr(), n, o[:B].data_ptr(),
                                           o[B:2 * B].data_ptr(), lse.data_ptr(), tr.dy[:B].data_ptr(), n, tr.dy[B:].data_ptr(), n,
                                           tr.loss_ws.data_ptr(), tr.loss_ws.numel(), st), "fwd_train")
    between()
    _lib.check(lib.clica_lp_loss_bwd_sym_train(C.byref(tr.desc), y1.data_ptr(), n, y1.data_ptr(), n, lse.data_ptr(), lse.data_ptr(),
                                               tr.dy[:B].data_ptr(), n, o[3 * B:].data_ptr(), tr.step_dev.data_ptr() if tr.early_tick else None,
                                               tr.loss_ws.data_ptr(), tr.loss_ws.numel(), _lib.stream_ptr()), "bwd_sym_train")
    tr._ticked = tr.early_tick


def body(variant):
    main, side = torch.cuda.current_stream(), tr.side_stream
    tr._packed_current = False
    side.wait_stream(main)
    with torch.cuda.stream(side):
        tr.pack()
    tr.sample()
    main.wait_stream(side)
    tr.forward()
    if variant == "base":
        tr.loss_forward_backward()
        tr.backward_chain(tr.dy)
    elif variant == "split":
        def fork():
            side.wait_stream(main)
            with torch.cuda.stream(side):
                chain_rows(tr.dy, B, 2 * B)
        loss_two_calls(fork)
        chain_rows(tr.dy, 0, B)
        main.wait_stream(side)
    else:
        loss_two_calls(lambda: None)
        chain_rows(tr.dy, B, 2 * B)
        chain_rows(tr.dy, 0, B)
    tr.weight_grads(tr.dy)
    tr.optimizer_step()


def measure(variant, reps=400):
    for _ in range(3):
        body(variant)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        g.capture_begin()
        body(variant)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(cap)
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


ref = None
for rnd in range(2):
    for v in ("base", "split", "serial"):
        us = measure(v)
        print(f"round {rnd} {v:7s} {us:8.1f} us/step   loss {float(tr.loss_out[3 * B]):.4f}", flush=True)
