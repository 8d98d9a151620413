#!/usr/bin/env python3
"""One wide layer (BASELINE config 3: M = 12288 rows, 2000 x 2000): fp32-MFMA per-layer kernels (linear.hip) against the split-bf16
GEMMs with fused epilogue (wgrad_split.hip: gemm_split_k).    python tools/split_gemm_bench.py [M N K]    (GPU box)
CLICA_SPLIT_GEMM_TILE=0/1/2 forces the tile shape of the split kernel (128 x 256 / 256 x 128 / 256 x 256)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cl_ica_amd import ops

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (12288, 2000, 2000)
dev = "cuda"
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
dz = torch.randn(M, N, device=dev)
xT = ops.mlp_planes_from_f32_t(x); wT = ops.mlp_planes_from_f32_t(w); wN = ops.mlp_planes_from_f32(w, False)
dzT = ops.mlp_planes_from_f32_t(dz)
yT = ops.mlp_planes_alloc(N, M, False, dev); yN = ops.mlp_planes_alloc(M, N, True, dev); y = torch.empty(M, N, device=dev)
dxT = ops.mlp_planes_alloc(K, M, False, dev); dxN = ops.mlp_planes_alloc(M, K, False, dev); dx = torch.empty(M, K, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


gf = 2.0 * M * N * K / 1e9
rows = [
    ("fp32 MFMA forward (linear_fwd)", lambda: ops.linear_fwd(x, w, b, leaky=True, slope=0.01, out=y)),
    ("split forward -> T + N planes", lambda: ops.linear_split_fwd(xT, wT, b, M, N, K, True, 0.01, yT=yT, yN=yN)),
    ("split forward -> T planes only", lambda: ops.linear_split_fwd(xT, wT, b, M, N, K, True, 0.01, yT=yT)),
    ("split forward -> fp32 only", lambda: ops.linear_split_fwd(xT, wT, b, M, N, K, True, 0.01, y=y)),
    ("fp32 MFMA dgrad (linear_dgrad)", lambda: ops.linear_dgrad(dz, w, x, 0.01, out=dx)),
    ("split dgrad -> T + N planes", lambda: ops.linear_split_dgrad(dzT, wN, xT, 0.01, M, N, K, dxT=dxT, dxN=dxN)),
    ("fp32 -> T planes (conversion)", lambda: ops.mlp_planes_from_f32_t(x, out=xT)),
    ("fp32 -> N planes (conversion)", lambda: ops.mlp_planes_from_f32(dz, False, out=dxN) if N == K else None),
]
print(f"M = {M}, N = {N}, K = {K}: {gf:.1f} GFLOP per GEMM; tile = {os.environ.get('CLICA_SPLIT_GEMM_TILE', 'auto')}")
for name, fn in rows:
    us = timeit(fn)
    print(f"  {name:34s} {us:8.1f} us   {gf / us * 1e3:7.1f} TFLOP/s (fp32-equivalent)" if "conversion" not in name else f"  {name:34s} {us:8.1f} us")
