#!/usr/bin/env python3
"""Kernel-time probe of the tiled Lp-InfoNCE forward/backward (graph replay between events)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import _lib

def run(B, B3, n, p, reps=20):
    lib = _lib.load()
    d = _lib.LpLossDesc(B=B, B3=B3, n=n, p=float(p), tau=1.0, alpha=0.5, compat=1, pow=1)
    fb, bb = C.c_size_t(), C.c_size_t()
    lib.clica_lp_loss_workspace_bytes(C.byref(d), C.byref(fb), C.byref(bb))
    ws = torch.zeros(max(fb.value, bb.value) + (64 << 20), dtype=torch.uint8, device="cuda")
    z1 = torch.randn(B, n, device="cuda") * 0.5; z2 = z1 + 0.05 * torch.randn_like(z1); z3 = torch.randn(B3, n, device="cuda") * 0.5
    o = torch.empty(3 * B + 3, device="cuda"); dz = torch.empty(2 * B + B3, n, device="cuda")
    rg = torch.empty(B, n, device="cuda"); RG = os.environ.get("LOSS_ROWGRAD", "1") == "1"
    def fwd():
        lib.clica_lp_loss_fwd(C.byref(d), z1.data_ptr(), n, z2.data_ptr(), n, z3.data_ptr(), n, o[:B].data_ptr(), o[B:2*B].data_ptr(),
                              o[2*B:3*B].data_ptr(), o[3*B:].data_ptr(), (rg.data_ptr() if RG else None), n, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    def bwd():
        lib.clica_lp_loss_bwd(C.byref(d), z1.data_ptr(), n, z2.data_ptr(), n, z3.data_ptr(), n, o[2*B:3*B].data_ptr(), (rg.data_ptr() if RG else None), n, None, None, None, None,
                              dz[:B].data_ptr(), n, dz[B:2*B].data_ptr(), n, dz[2*B:].data_ptr(), n, 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    res = []
    for fn in (fwd, bwd):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): g.replay()
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) * 1e3 / reps)
    return res

if __name__ == "__main__":
    import subprocess
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        for (B, B3, n, p) in ((6144, 6144, 10, 2), (6144, 6144, 10, 1), (6144, 49152, 10, 2)):
            f, b = run(B, B3, n, p)
            print(f"WG/CU={os.environ.get('CLICA_LP_WG_PER_CU','8')} B={B} B3={B3} n={n} p={p}: fwd {f:8.1f} us ({B*B3/f/1e3:7.1f} Gpair/s)  bwd {b:8.1f} us")
    else:
        for rgflag in ("0", "1"):
            print("row gradient in the forward sweep:", rgflag)
            env = dict(os.environ, LOSS_ROWGRAD=rgflag)
            subprocess.run([sys.executable, __file__, "one"], env=env)
