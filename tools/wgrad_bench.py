#!/usr/bin/env python3
"""Weight gradients of the n=10 encoder: grouped launch (clica_mlp_wgrad) vs seven per-layer launches, graph replay.
CLICA_WGRAD_GROUP_SPLITS=<s> overrides the planner's common split count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import ops
from tools.mlp_bench import replay_time

dims = [10, 100, 500, 500, 500, 500, 100, 10]
M = 12288
torch.manual_seed(0)
xs = [torch.randn(M, dims[l], device="cuda") for l in range(7)]
dzs = [torch.randn(M, dims[l + 1], device="cuda") for l in range(7)]
dWs = [torch.empty(dims[l + 1], dims[l], device="cuda") for l in range(7)]
dbs = [torch.empty(dims[l + 1], device="cuda") for l in range(7)]
flops = 2.0 * M * sum(dims[i] * dims[i + 1] for i in range(7))
ws = ops.mlp_wgrad_workspace(M, [(dims[l + 1], dims[l]) for l in range(7)], "cuda")
t = replay_time(lambda: ops.mlp_wgrad(dzs, xs, dWs, dbs, ws=ws))
print(f"grouped (splits env={os.environ.get('CLICA_WGRAD_GROUP_SPLITS','plan')}, ws {ws.numel()/1e6:.1f} MB): {t:7.1f} us  {flops/t/1e6:6.1f} TFLOP/s")
if "--per-layer" in sys.argv:
    def per():
        for l in range(7):
            ops.linear_wgrad(dzs[l], xs[l], dW=dWs[l], db=dbs[l])
    t = replay_time(per)
    print(f"7 per-layer launches: {t:7.1f} us  {flops/t/1e6:6.1f} TFLOP/s")
