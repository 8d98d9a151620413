#!/usr/bin/env python3
"""SimCLRLoss (losses.py:162-202) forward + backward time: pair sweep on the vector ALU against the MFMA path (logit matrix + three
fp32-MFMA GEMMs, csrc/lp_loss.hip "SimCLRLoss on the matrix cores"), over the row width n.
    python tools/simclr_bench.py [B]      (GPU box)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cl_ica_amd import _lib
from cl_ica_amd.losses import SimCLRLoss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _lib.load()
L = SimCLRLoss(normalize=True, tau=0.5, alpha=0.5)
print(f"SimCLRLoss fwd+bwd, B = B3 = {B}, normalize, times in us (median of 20)")
for n in (32, 64, 128, 256, 512):
    torch.manual_seed(0)
    z = [torch.randn(B, n, device="cuda").requires_grad_(True) for _ in range(3)]
    row = [f"n = {n:4d}"]
    for path in ("0", "1"):
        lib.clica_set_tuning(b"dot_mfma", int(path))
        ts = []
        for it in range(25):
            for t in z:
                t.grad = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tot, _, _ = L(None, None, None, *z)
            tot.backward()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        ts = sorted(ts[5:])
        row.append(f"{'mfma' if path == '1' else 'sweep'} {ts[len(ts) // 2]:9.1f}")
    flops = 3 * 2 * B * B * n          # S (twice: forward and backward recompute it), W z3, W^T z1 -> 4 GEMMs; count the 3 distinct
    print("   ".join(row), f"   (one GEMM = {2 * B * B * n / 1e9:.2f} GFLOP)")
os.environ.pop("CLICA_DOT_MFMA", None)
lib.clica_reload_env()
