"""Time clica_nn_search at the 3DIdent table size (250 000 x 10) for a batch of 2 x 1024 queries."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import ops  # noqa: E402


def main():
    torch.manual_seed(0)
    for (N, Q, n, k) in ((250000, 1024, 10, 1), (250000, 1024, 10, 2), (250000, 2048, 10, 2), (1000000, 2048, 10, 2)):
        tab = torch.rand(N, n, device="cuda") * 2 - 1
        qry = torch.rand(Q, n, device="cuda") * 2 - 1
        for _ in range(5):
            ops.nn_search(tab, qry, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.nn_search(tab, qry, k)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"N={N} Q={Q} n={n} k={k}: {us:8.1f} us  {N * Q / us / 1e6:7.3f} Tpair/s  (table {N * n * 4 / 1e6:.0f} MB)")


if __name__ == "__main__":
    main()
