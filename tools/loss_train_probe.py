#!/usr/bin/env python3
"""Training-path loss kernels (clica_lp_loss_fwd_train / clica_lp_loss_bwd_sym_train) timed by graph replay, for planner sweeps:
CLICA_LP_WG_PER_CU_FWD / CLICA_LP_WG_PER_CU force "about k workgroups per CU".  usage: loss_train_probe.py [one]"""
import ctypes as C, os, subprocess, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one():
    from cl_ica_amd import _lib
    lib = _lib.load()
    for (B, B3, n, p) in ((6144, 6144, 10, 2), (6144, 49152, 10, 2), (6144, 6144, 40, 1)):
        d = _lib.LpLossDesc(B=B, B3=B3, n=n, p=float(p), tau=1.0, alpha=0.5, compat=1, pow=1)
        nb = C.c_size_t(); lib.clica_lp_loss_train_workspace_bytes(C.byref(d), C.byref(nb))
        ws = torch.zeros(nb.value, dtype=torch.uint8, device="cuda")
        pool = torch.rand(B3, n, device="cuda"); z1 = pool[:B]; z2 = (z1 + 0.05 * torch.randn_like(z1)).clamp(0, 1)
        o = torch.empty(3 * B + 3, device="cuda"); dy = torch.empty(2 * B, n, device="cuda"); lse_all = torch.zeros(B3, device="cuda")
        st = _lib.stream_ptr()
        def fwd():
            lib.clica_lp_loss_fwd_train(C.byref(d), z1.data_ptr(), n, z2.data_ptr(), n, pool.data_ptr(), n, o[:B].data_ptr(), o[B:2 * B].data_ptr(),
                                        o[2 * B:3 * B].data_ptr(), dy[:B].data_ptr(), n, dy[B:].data_ptr(), n, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        def bwd():
            pl = o[2 * B:3 * B] if B3 == B else lse_all
            lib.clica_lp_loss_bwd_sym_train(C.byref(d), z1.data_ptr(), n, pool.data_ptr(), n, o[2 * B:3 * B].data_ptr(), pl.data_ptr(),
                                            dy[:B].data_ptr(), n, o[3 * B:].data_ptr(), None, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        res = []
        for fn in (fwd, bwd):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            g.replay(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                s.record()
                for _ in range(30): g.replay()
                e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) * 1e3 / 30)
            res.append(best)
        print(f"  B={B} B3={B3} n={n} p={p}: fwd_train {res[0]:7.1f} us  bwd_sym_train {res[1]:7.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for k in os.environ.get("SWEEP", "0,2,3,4,5,6").split(","):
            env = dict(os.environ)
            if k != "0":
                env["CLICA_LP_WG_PER_CU"] = k; env["CLICA_LP_WG_PER_CU_FWD"] = k
            print(f"WG/CU forced = {k} (0 = planner)  lib = {os.environ.get('CLICA_LIB', 'default')}", flush=True)
            subprocess.run([sys.executable, __file__, "one"], env=env)
