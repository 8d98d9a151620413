#!/usr/bin/env python3
"""Phase timing inside the split-bf16 weight-gradient GEMM (-DCLICA_WSPLIT_TRACE build of csrc/wgrad_split.hip): per work
item prologue / step loop / slab epilogue, and one steady-state 16-row step (t = 30) in detail.
    make -C cl_ica_amd/csrc variant SRC=wgrad_split.hip TAG=wstrace EXTRA=-DCLICA_WSPLIT_TRACE && python tools/wsplit_trace.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLICA_LIB", os.path.join(ROOT, "cl_ica_amd/lib/libclica_hip_wstrace.so"))
from cl_ica_amd import _lib, ops, encoders
from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
lib = _lib.load()
lib.clica_debug_wsplit_trace.argtypes = [ctypes.c_void_p]
n, B = 10, 6144
torch.manual_seed(0)
f = encoders.get_mlp(n, n, [n * 10, n * 50, n * 50, n * 50, n * 50, n * 10]).to("cuda")
gW = torch.randn(3, n, n, device="cuda") / n ** 0.5
tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=1e-4, device="cuda", split_bf16=True)
assert tr.split_wgrad
for _ in range(30):
    tr.step()
torch.cuda.synchronize()
NWG = 512
buf = torch.zeros(NWG * 8 * 16, dtype=torch.int64, device="cuda")
assert lib.clica_debug_wsplit_trace(buf.data_ptr()) == 0
tr.step(); torch.cuda.synchronize()
lib.clica_debug_wsplit_trace(None)
t = buf.cpu().numpy().reshape(NWG, 8, 16).astype(np.float64)
t = t[t[:, 0, 0] > 0]
print("work items:", len(t))
t0 = t[:, :, 0].min()
print(f"kernel span (first entry -> last exit) {t[:, :, 3].max() - t0:.0f} cycles; start spread {np.percentile(t[:, :, 0].min(1) - t0, [0, 50, 100]).round().tolist()}")
names = {5: "first half's MFMAs + reads issued", 6: "vmcnt wait done", 7: "barrier passed", 8: "second half + DMA requests issued", 10: "next step starts"}
for label, sel in (("256 x 256 items", t[:, 0, 11] >= 1000), ("256 x 128 / 128 x 256 items", t[:, 0, 11] < 1000)):
    u = t[sel]
    if not len(u):
        continue
    nt = int(np.median(u[:, 0, 11] % 1000)) if label.startswith("256 x 256") else -1
    pro, loop, epi = u[:, :, 1] - u[:, :, 0], u[:, :, 2] - u[:, :, 1], u[:, :, 3] - u[:, :, 2]
    print(f"== {label}: {len(u)}  (steps per item {nt if nt >= 0 else 'n/a'})")
    print(f"  median cycles per wave: prologue {np.median(pro):.0f}  step loop {np.median(loop):.0f}  epilogue {np.median(epi):.0f};  item exit - kernel start: {np.percentile(u[:, :, 3].max(1) - t0, [0, 50, 100]).round().tolist()}")
    if nt > 0:
        print(f"  loop cycles per step {np.median(loop) / nt:.0f}  (MFMA work per SIMD and step: 2 waves x 24 x 32 = 1536 in f16x2, 3072 in bf16x3)")
    print("  step loop by wave id (median):", np.round(np.median(loop, 0)).astype(int).tolist())
    print("  step 30, cycles since its start (median by wave id):")
    for k, nm in names.items():
        print(f"    {nm:34s}", np.round(np.median(u[:, :, k] - u[:, :, 4], 0)).astype(int).tolist())
