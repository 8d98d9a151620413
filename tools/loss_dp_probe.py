#!/usr/bin/env python3
"""Loss forward + symmetric backward at the pool sizes of R-rank data parallelism (B3 = R*B, pool contains the local rows)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import _lib
from tools.mlp_bench import replay_time

lib = _lib.load()
B, n, p = 6144, 10, 2.0
for R in (1, 2, 4, 8):
    B3 = R * B
    d = _lib.LpLossDesc(B=B, B3=B3, n=n, p=p, tau=1.0, alpha=0.5, compat=1, pow=1)
    fb, bb = C.c_size_t(), C.c_size_t()
    lib.clica_lp_loss_workspace_bytes(C.byref(d), C.byref(fb), C.byref(bb))
    ws = torch.zeros(max(fb.value, bb.value), dtype=torch.uint8, device="cuda")
    pool = torch.rand(B3, n, device="cuda")
    z1 = pool[:B]; z2 = (z1 + 0.05 * torch.randn_like(z1)).contiguous()
    o = torch.empty(3 * B + 3, device="cuda"); dz = torch.empty(2 * B, n, device="cuda")
    lse_pool = torch.zeros(B3, device="cuda")
    st = _lib.stream_ptr
    def fwd():
        _lib.check(lib.clica_lp_loss_fwd(C.byref(d), z1.data_ptr(), n, z2.data_ptr(), n, pool.data_ptr(), n, o[:B].data_ptr(), o[B:2*B].data_ptr(),
                                         o[2*B:3*B].data_ptr(), o[3*B:].data_ptr(), None, 0, ws.data_ptr(), ws.numel(), st()), "fwd")
    def bwd():
        _lib.check(lib.clica_lp_loss_bwd_sym(C.byref(d), z1.data_ptr(), n, z2.data_ptr(), n, pool.data_ptr(), n, o[2*B:3*B].data_ptr(),
                                             (o[2*B:3*B] if R == 1 else lse_pool).data_ptr(), None, None, None, dz[:B].data_ptr(), n,
                                             dz[B:].data_ptr(), n, ws.data_ptr(), ws.numel(), st()), "bwd")
    fwd(); lse_pool[:B] = o[2*B:3*B]; lse_pool[B:] = o[2*B:3*B].repeat(R)[:B3 - B] if R > 1 else lse_pool[B:]
    tf, tb = replay_time(fwd), replay_time(bwd)
    print(f"R={R} B3={B3}: fwd {tf:7.1f} us ({B*B3/tf/1e6:5.2f} Tpair/s)  bwd_sym {tb:7.1f} us ({B*B3/tb/1e6:5.2f} Tpair/s)  total {tf+tb:7.1f} us", flush=True)
