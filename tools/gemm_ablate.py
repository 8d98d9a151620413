#!/usr/bin/env python3
"""Ablation probe for the forward GEMM main loop (results are WRONG by construction when ablated)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import ops
from gemm_bench import timeit
M, N, K = 12288, 500, 500
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
for cfg in (0, 5):
    os.environ["CLICA_GEMM_CFG_FWD"] = str(cfg)
    for ab, name in ((0, "full"), (1, "no global loads"), (3, "no gl loads, no lds stores"), (7, "mfma+barrier only"), (15, "mfma only"), (8, "no barrier"), (4, "no frag reads")):
        os.environ["CLICA_GEMM_ABLATE"] = str(ab)
        t = timeit(lambda: ops.linear_fwd(x, w, b, True), reps=30)
        print(f"cfg {cfg} ablate {ab:2d} ({name:28s}): {t*1e6:7.1f} us")
for K2 in (32, 100, 250, 500, 1000, 2000):
    os.environ["CLICA_GEMM_ABLATE"] = "0"; os.environ["CLICA_GEMM_CFG_FWD"] = "0"
    x2 = torch.randn(M, K2, device="cuda"); w2 = torch.randn(N, K2, device="cuda")
    t = timeit(lambda: ops.linear_fwd(x2, w2, b, True), reps=30)
    print(f"K={K2}: {t*1e6:7.1f} us  {2.0*M*N*K2/t/1e12:6.1f} TF")
