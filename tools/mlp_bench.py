#!/usr/bin/env python3
"""Forward of the n=10 encoder stack: one-launch fused kernel vs seven per-layer GEMM launches (graph replay)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import ops

def replay_time(fn, reps=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps

def main():
    dims = [10, 100, 500, 500, 500, 500, 100, 10]
    M = 12288
    torch.manual_seed(0)
    Ws = [torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5 for i in range(7)]
    bs = [torch.randn(dims[i + 1], device="cuda") * 0.1 for i in range(7)]
    x = torch.randn(M, 10, device="cuda")
    outs = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    def fused(): ops.mlp_fwd(x, Ws, bs, outs, 0.01)
    def layered():
        cur = x
        for l in range(7):
            ops.linear_fwd(cur, Ws[l], bs[l], leaky=(l < 6), slope=0.01, out=outs[l]); cur = outs[l]
    flops = 2.0 * M * sum(dims[i] * dims[i + 1] for i in range(7))
    packed = ops.mlp_pack_weights(Ws)
    def fused_packed(): ops.mlp_fwd(x, Ws, bs, outs, 0.01, packed=packed)
    def pack(): ops.mlp_pack_weights(Ws, packed)
    for name, fn in (("fused (direct W)", fused), ("fused (packed W)", fused_packed)):
        tf = replay_time(fn)
        print(f"{name}: {tf:7.1f} us  {flops/tf/1e6:6.1f} TFLOP/s")
    print(f"pack kernel: {replay_time(pack):6.1f} us")
    tl = replay_time(layered)
    print(f"7 layers: {tl:7.1f} us  {flops/tl/1e6:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
