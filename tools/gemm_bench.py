#!/usr/bin/env python3
"""Micro-benchmark of the fused Linear GEMM kernels per tile configuration (tuning aid).
Interleaves configurations in one process (A/B within a probe) and checks each against torch."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import _lib, ops  # noqa: E402

SHAPES = [(12288, 500, 500), (12288, 100, 500), (12288, 500, 100), (12288, 100, 10), (12288, 10, 100)]
if len(sys.argv) > 1 and sys.argv[1] == "n40":
    SHAPES = [(12288, 2000, 2000), (12288, 400, 2000), (12288, 2000, 400)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def main():
    torch.manual_seed(0)
    for (M, N, K) in SHAPES:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda"); dy = torch.randn(M, N, device="cuda"); xa = torch.randn(M, K, device="cuda")
        ref_y = torch.nn.functional.leaky_relu(x.double() @ w.double().T + b.double(), 0.01)
        ref_dx = (dy.double() @ w.double()) * torch.where(xa > 0, 1.0, 0.01).double()
        ref_dw = dy.double().T @ x.double()
        fl = 2.0 * M * N * K
        print(f"--- M={M} N={N} K={K} ({fl/1e9:.2f} GFLOP)")
        for cfg in ("auto", 0, 1, 2, 3, 4, 5):
            for key in (b"gemm_cfg_fwd", b"gemm_cfg_dgrad", b"gemm_cfg_wgrad"):
                _lib.load().clica_set_tuning(key, -1 if cfg == "auto" else int(cfg))
            y = ops.linear_fwd(x, w, b, True); dx = ops.linear_dgrad(dy, w, xa); dw, db = ops.linear_wgrad(dy, x)
            errs = [float((y - ref_y).abs().max() / ref_y.abs().max()), float((dx - ref_dx).abs().max() / ref_dx.abs().max()),
                    float((dw - ref_dw).abs().max() / ref_dw.abs().max())]
            tf = timeit(lambda: ops.linear_fwd(x, w, b, True)); td = timeit(lambda: ops.linear_dgrad(dy, w, xa))
            tw = timeit(lambda: ops.linear_wgrad(dy, x))
            print(f"cfg {cfg!s:>4}: fwd {tf*1e6:7.1f}us {fl/tf/1e12:6.1f}TF | dgrad {td*1e6:7.1f}us {fl/td/1e12:6.1f}TF | "
                  f"wgrad(+reduce) {tw*1e6:7.1f}us {fl/tw/1e12:6.1f}TF | err {max(errs):.1e}")
    for var in ("CLICA_GEMM_CFG_FWD", "CLICA_GEMM_CFG_DGRAD", "CLICA_GEMM_CFG_WGRAD"):
        os.environ.pop(var, None)


if __name__ == "__main__":
    main()
