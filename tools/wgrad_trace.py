#!/usr/bin/env python3
"""Phase timing inside the grouped weight-gradient GEMM (-DCLICA_WGRAD_TRACE build): per work item prologue (first three tile
DMAs landed), k-loop, epilogue (slab store), and when each item started relative to the launch.
    make -C cl_ica_amd/csrc trace && python tools/wgrad_trace.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLICA_LIB", os.path.join(ROOT, "cl_ica_amd/lib/libclica_hip_trace.so"))
from cl_ica_amd import _lib, ops
lib = _lib.load()
lib.clica_debug_wgrad_trace.argtypes = [ctypes.c_void_p]
dims = [10, 100, 500, 500, 500, 500, 100, 10]; M = 12288
idx = [1, 2, 3, 4, 5]
torch.manual_seed(0)
xs = [torch.randn(M, dims[l], device="cuda") for l in idx]
dzs = [torch.randn(M, dims[l + 1], device="cuda") for l in idx]
dWs = [torch.empty(dims[l + 1], dims[l], device="cuda") for l in idx]
dbs = [torch.empty(dims[l + 1], device="cuda") for l in idx]
ws = ops.mlp_wgrad_workspace(M, [(dims[l + 1], dims[l]) for l in idx], "cuda")
run = lambda: ops.mlp_wgrad(dzs, xs, dWs, dbs, ws=ws)
for _ in range(20): run()
torch.cuda.synchronize()
NWG = 1024
buf = torch.zeros(NWG * 8 * 8, dtype=torch.int64, device="cuda")
assert lib.clica_debug_wgrad_trace(buf.data_ptr()) == 0
run(); torch.cuda.synchronize()
lib.clica_debug_wgrad_trace(None)
t = buf.cpu().numpy().reshape(NWG, 8, 8).astype(np.float64)
used = t[:, 0, 0] > 0
t = t[used]
print("work items:", int(used.sum()))
t0 = t[:, :, 0].min()
start = t[:, :, 0].min(1) - t0
pro, loop, epi = t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2]
end = t[:, :, 3].max(1) - t0
print(f"median cycles per wave: prologue {np.median(pro):.0f}  k-loop {np.median(loop):.0f}  epilogue {np.median(epi):.0f}   (ideal k-loop of a 256 x 128 item: 43 tiles x 8192 = 352256 per SIMD pair of waves, i.e. 4096 MFMA cycles per wave and tile)")
print(f"k-loop by wave id (median): {np.round(np.median(loop, 0)).astype(int).tolist()}")
order = np.argsort(start)
print("item start times (cycles, sorted): first round", np.round(np.percentile(start, [0, 25, 49]), 0).tolist(), " second round", np.round(np.percentile(start, [51, 75, 100]), 0).tolist())
print(f"kernel span: {end.max():.0f} cycles; last first-round end {np.sort(end)[min(255, len(end) - 1)]:.0f}")
# one tile in detail (t = 20): stamp4 = tile start, 5 = k-step 0's MFMAs issued, 6 = past wait + barrier, 7 = k-step 1 start (DMA issued)
fine = t[:, :, 5:8] - t[:, :, 4:5]
print("tile 20, cycles since its start (median by wave id):")
for nm, k in (("k-step 0 MFMAs issued", 0), ("past vmcnt wait + barrier", 1), ("k-step 1 starts (4 DMAs issued)", 2)):
    print(f"  {nm:34s}", np.round(np.median(fine[:, :, k], 0)).astype(int).tolist())
print("  whole tile 20 (start of 21 - start of 20):", np.round(np.median(t[:, :, 0], 0)).astype(int).tolist(), " [slot 0 overwritten: item start unavailable in this build]")
