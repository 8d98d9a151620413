import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import ops
dims = [10, 100, 500, 500, 500, 500, 100, 10]; M = 12288
torch.manual_seed(0)
idx = [0, 6]
xs = [torch.randn(M, dims[l], device="cuda") for l in idx]
dzs = [torch.randn(M, dims[l + 1], device="cuda") for l in idx]
dWs = [torch.empty(dims[l + 1], dims[l], device="cuda") for l in idx]
dbs = [torch.empty(dims[l + 1], device="cuda") for l in idx]
ws = ops.mlp_wgrad_workspace(M, [(dims[l + 1], dims[l]) for l in idx], "cuda")
for _ in range(20):
    ops.mlp_wgrad(dzs, xs, dWs, dbs, ws=ws)
torch.cuda.synchronize()
