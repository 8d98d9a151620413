#!/usr/bin/env python3
"""Where does mlp_fwd_k spend its time?  Runs the n = 10 forward stack on a -DCLICA_FMLP_TRACE build of the library (s_memtime stamps per
workgroup / wave / layer / phase) and prints, per layer, the median cycles of: k-loop, wait at the post-GEMM barrier, epilogue,
wait at the post-epilogue barrier, activation-store issue.
    make -C cl_ica_amd/csrc trace      # builds cl_ica_amd/lib/libclica_hip_trace.so (here or on the GPU box)
    CLICA_LIB=cl_ica_amd/lib/libclica_hip_trace.so python tools/fmlp_trace.py [bwd]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLICA_LIB", os.path.join(ROOT, "cl_ica_amd/lib/libclica_hip_trace.so"))
from cl_ica_amd import _lib, ops
lib = _lib.load()
lib.clica_debug_fmlp_trace.argtypes = [ctypes.c_void_p]
dims = [10, 100, 500, 500, 500, 500, 100, 10]
if len(sys.argv) > 2:
    dims = [int(v) for v in sys.argv[2].split(",")]
M = 12288; L = len(dims) - 1
torch.manual_seed(0)
Ws = [torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5 for i in range(L)]
bs = [torch.randn(dims[i + 1], device="cuda") * 0.1 for i in range(L)]
x = torch.randn(M, dims[0], device="cuda")
outs = [torch.empty(M, d, device="cuda") for d in dims[1:]]
packed, packed_t = ops.mlp_pack_both(Ws)
masks = ops.mlp_signmask_alloc(M, L - 1, "cuda") + [None]
WG, WAVES, MAXL = (M + 47) // 48, 8, 8
buf = torch.zeros(WG * WAVES * (MAXL + 1) * 8, dtype=torch.int64, device="cuda")
bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
def run():
    if not bwd:
        ops.mlp_fwd(x, Ws, bs, outs, 0.01, packed=packed, signmasks=masks)
    else:
        chain = list(range(L - 1, 0, -1))
        ops.mlp_dgrad_chain(dy, [Ws[l] for l in chain], packed_t, [outs[l - 1] for l in chain], dz, 0.01, masks_chain=[masks[l - 1] for l in chain])
ops.mlp_fwd(x, Ws, bs, outs, 0.01, packed=packed, signmasks=masks)
dy = torch.randn(M, dims[-1], device="cuda"); dz = [torch.empty(M, dims[l], device="cuda") for l in range(L - 1, 0, -1)]
for _ in range(5): run()
torch.cuda.synchronize()
assert lib.clica_debug_fmlp_trace(buf.data_ptr()) == 0
run(); torch.cuda.synchronize()
lib.clica_debug_fmlp_trace(None)
t = buf.cpu().numpy().reshape(WG, WAVES, MAXL + 1, 8).astype(np.float64)
nl = L if not bwd else L - 1
start = t[:, :, MAXL, 0]
print(f"{'layer':>5} {'kloop':>8} {'bar1':>7} {'epi':>7} {'bar2':>7} {'store':>7} {'total':>8}   (median cycles per wave; kloop max over waves)")
tot = 0
for l in range(nl):
    a = t[:, :, l, :]
    k, b1, ep, b2, stc = a[..., 1] - a[..., 0], a[..., 2] - a[..., 1], a[..., 3] - a[..., 2], a[..., 4] - a[..., 3], a[..., 5] - a[..., 4]
    full = a[..., 5] - a[..., 0]
    print(f"{l:5d} {np.median(k):8.0f} {np.median(b1):7.0f} {np.median(ep):7.0f} {np.median(b2):7.0f} {np.median(stc):7.0f} {np.median(full):8.0f}   kloop max/WG {np.median(k.max(1)):8.0f}")
    tot += np.median(full)
end = t[:, :, nl - 1, 5]
print("prologue (start -> layer 0):", np.median(t[:, :, 0, 0] - start), " whole WG:", np.median((end - start).max(1)), " sum of layer medians:", tot)
print("kernel span (max end - min start):", end.max() - start.min(), "cycles")
for l in (2, 3):
    a = t[:, :, l, :]
    k = a[..., 1] - a[..., 0]
    print(f"layer {l}: kloop median by wave id:", np.round(np.median(k, 0)).astype(int).tolist())
    print(f"layer {l}: layer-start (stamp0 - WG min) by wave id:", np.round(np.median(a[..., 0] - a[..., 0].min(1, keepdims=True), 0)).astype(int).tolist())
    print(f"layer {l}: kloop END (stamp1 - WG min start) by wave id:", np.round(np.median(a[..., 1] - a[..., 0].min(1, keepdims=True), 0)).astype(int).tolist())
