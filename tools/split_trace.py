#!/usr/bin/env python3
"""Phase timing inside the split-bf16 whole-stack kernel (-DCLICA_SPLIT_TRACE build of csrc/fused_mlp.hip): per layer and wave the
k-loop, the wait at the barrier behind it, the epilogue (bias / activation / split / LDS + HBM stores) and the barrier behind that.
    make -C cl_ica_amd/csrc variant SRC=fused_mlp.hip TAG=strace EXTRA=-DCLICA_SPLIT_TRACE && python tools/split_trace.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLICA_LIB", os.path.join(ROOT, "cl_ica_amd/lib/libclica_hip_strace.so"))
from cl_ica_amd import _lib, encoders
from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
lib = _lib.load()
lib.clica_debug_split_trace.argtypes = [ctypes.c_void_p]
n, B = 10, 6144
torch.manual_seed(0)
f = encoders.get_mlp(n, n, [n * 10, n * 50, n * 50, n * 50, n * 50, n * 10]).to("cuda")
gW = torch.randn(3, n, n, device="cuda") / n ** 0.5
tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=1e-4, device="cuda", split_bf16=True)
for _ in range(30):
    tr.step()
torch.cuda.synchronize()
NWG, WAVES, MAXL = 256, 8, 8
for which in ("forward", "chain"):
    NS = 8
    buf = torch.zeros(NWG * WAVES * MAXL * NS, dtype=torch.int64, device="cuda")
    tr._packed_current = False
    tr.sample(); tr.pack()
    if which == "forward":
        if os.environ.get("TRACE_WARM_ICACHE") == "1":      # the same launch right before the traced one: warm instruction cache / TLBs
            tr.forward()
        assert lib.clica_debug_split_trace(buf.data_ptr()) == 0
        tr.forward(); torch.cuda.synchronize()
        lib.clica_debug_split_trace(None)
        tr.loss_forward_backward()
    else:
        tr.forward(); tr.loss_forward_backward(); torch.cuda.synchronize()
        assert lib.clica_debug_split_trace(buf.data_ptr()) == 0
        tr.backward_chain(tr.dy); torch.cuda.synchronize()
        lib.clica_debug_split_trace(None)
    t = buf.cpu().numpy().reshape(NWG, WAVES, MAXL, NS).astype(np.float64)
    L = 7 if which == "forward" else 6
    print(f"== {which}: cycles per layer (median over workgroups); ideal MFMA cycles of a 500 x 500 layer: 16 iterations x 72 x 16 = 18432 per wave, 36864 per SIMD pair")
    t0 = t[:, :, 0, 0].min(1, keepdims=True)
    for l in range(L):
        kl = t[:, :, l, 1] - t[:, :, l, 0]; b1 = t[:, :, l, 2] - t[:, :, l, 1]; ep = t[:, :, l, 3] - t[:, :, l, 2]
        nxt = (t[:, :, l + 1, 0] - t[:, :, l, 3]) if l + 1 < L else np.zeros_like(kl)
        print(f"  layer {l}: k-loop by wave {np.round(np.median(kl, 0)).astype(int).tolist()}  barrier wait {np.round(np.median(b1, 0)).astype(int).tolist()}"
              f"  epilogue {int(np.median(ep))} [drain {np.round(np.median(t[:, :, l, 4] - t[:, :, l, 2], 0)).astype(int).tolist()} request {int(np.median(t[:, :, l, 5] - t[:, :, l, 4]))} body {np.round(np.median(t[:, :, l, 6] - t[:, :, l, 5], 0)).astype(int).tolist()} tail {int(np.median(t[:, :, l, 3] - t[:, :, l, 6]))}]  barrier 2 {int(np.median(nxt))}   layer total {int(np.median(t[:, :, l, 3].max(1) - t[:, :, l, 0].min(1)))}")
    e0 = t[:, :, MAXL - 1, 0].min(1)
    pro = [int(np.median(t[:, :, MAXL - 1, q].max(1) - e0)) for q in (1, 2, 3)]
    print(f"  prologue phases from the workgroup's first entry (last wave): loads consumed {pro[0]}, first barrier passed {pro[1]}, mixing done {pro[2]}, "
          f"first layer starts {int(np.median(t[:, :, 0, 0].min(1) - e0))}; entry skew of the eight waves {int(np.median(t[:, :, MAXL - 1, 0].max(1) - e0))}")
    print(f"  whole kernel (first stamp to last): {int(np.median(t[:, :, L - 1, 3].max(1) - t[:, :, 0, 0].min(1)))} cycles;  prologue (kernel entry -> "
          f"first layer: bias table, input staging / mixing net, zero fill): {int(np.median(t[:, :, 0, 0].min(1) - t[:, :, MAXL - 1, 0].min(1)))} cycles;  "
          f"spread of the workgroups' entry times: {int(t[:, :, MAXL - 1, 0].min(1).max() - t[:, :, MAXL - 1, 0].min(1).min())} cycles;  "
          f"first entry -> last exit over all workgroups: {int(t[:, :, L - 1, 3].max() - t[:, :, MAXL - 1, 0].min())} cycles")
