#!/usr/bin/env python3
"""Fused MLP forward: time vs number of 500x500 layers (slope = per-layer cost, intercept = fixed cost)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import ops
from tools.mlp_bench import replay_time
M = int(os.environ.get("M", 12288)); W = int(os.environ.get("W", 500))
torch.manual_seed(0)
x = torch.randn(M, W, device="cuda")
for L in (1, 2, 3, 4, 6, 8):
    Ws = [torch.randn(W, W, device="cuda") / W ** 0.5 for _ in range(L)]
    bs = [torch.randn(W, device="cuda") * 0.1 for _ in range(L)]
    outs = [torch.empty(M, W, device="cuda") for _ in range(L)]
    masks = ops.mlp_signmask_alloc(M, L, "cuda")
    if os.environ.get("SPLIT") == "1":
        packed = ops.mlp_pack_split_both(Ws + [Ws[-1]])[0] if L == 1 else ops.mlp_pack_split_both(Ws)[0]
        if L == 1:   # pack_both needs >= 2 layers: pack [W, W] and use the first
            pass
        t = replay_time(lambda: ops.mlp_fwd_split(x, Ws, bs, outs, packed, 0.01, signmasks=masks))
    else:
        packed = ops.mlp_pack_weights(Ws)
        t = replay_time(lambda: ops.mlp_fwd(x, Ws, bs, outs, 0.01, packed=packed, signmasks=masks))
    fl = 2.0 * M * W * W * L
    print(f"L={L}: {t:7.1f} us  {fl/t/1e6:6.1f} TFLOP/s   ({t/L:6.1f} us/layer)")
