#!/usr/bin/env python3
"""Where does the grouped weight-gradient launch lose time?  Times clica_mlp_wgrad (grouped GEMM + slab reduce, graph replay)
for the n = 10 stack with layer subsets and split counts.  usage: python tools/wgrad_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import ops


def replay_time(fn, reps=40):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        s.record()
        for _ in range(reps): g.replay()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    return best


def main():
    dims = [10, 100, 500, 500, 500, 500, 100, 10]
    M = 12288
    torch.manual_seed(0)
    xs = [torch.randn(M, dims[l], device="cuda") for l in range(7)]
    dzs = [torch.randn(M, dims[l + 1], device="cuda") for l in range(7)]
    dWs = [torch.empty(dims[l + 1], dims[l], device="cuda") for l in range(7)]
    dbs = [torch.empty(dims[l + 1], device="cuda") for l in range(7)]
    subsets = {"all7": list(range(7)), "no_L0_L6": [1, 2, 3, 4, 5], "square3": [2, 3, 4], "narrow(L1,L5)": [1, 5], "tiny(L0,L6)": [0, 6]}
    for name, idx in subsets.items():
        fl = 2.0 * M * sum(dims[l] * dims[l + 1] for l in idx)
        for sp in (os.environ.get("SPLITS", "0,8,9,10,13,16").split(",")):
            if sp == "0":
                os.environ.pop("CLICA_WGRAD_GROUP_SPLITS", None)
            else:
                os.environ["CLICA_WGRAD_GROUP_SPLITS"] = sp
            ws = ops.mlp_wgrad_workspace(M, [(dims[l + 1], dims[l]) for l in idx], "cuda")
            fn = lambda: ops.mlp_wgrad([dzs[l] for l in idx], [xs[l] for l in idx], [dWs[l] for l in idx], [dbs[l] for l in idx], ws=ws)
            t = replay_time(fn)
            print(f"{name:14s} splits={sp:>2s}: {t:7.1f} us  {fl / t / 1e6:6.1f} TFLOP/s  ws={ws.numel() / 1e6:.1f} MB", flush=True)
    os.environ.pop("CLICA_WGRAD_GROUP_SPLITS", None)


if __name__ == "__main__":
    main()
