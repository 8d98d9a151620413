"""Times every launch of the BetaVAE_H conv stack (clica_conv_*) at BASELINE configs[4]'s shapes, one kernel at a time (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cl_ica_amd._lib import load, check, stream_ptr, workspace

lib = load(); dev = torch.device("cuda")
N = int(os.environ.get("IMAGES", 2048))

def t_us(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

def rnd(*shape): return torch.randn(*shape, device=dev)
st = stream_ptr()
stages = [(1, 32, 32), (32, 32, 16), (32, 64, 8), (64, 64, 4)]   # (C, Cout, ho)
x = (torch.rand(N, 1, 64, 64, device=dev) < 0.3).float()
P = torch.empty(N * 1024, 16, device=dev)
print("im2col %.1f us" % t_us(lambda: check(lib.clica_conv_im2col_k4s2(x.data_ptr(), N, 1, 64, 64, P.data_ptr(), st), "im2col")))
S2 = torch.zeros(N * 17 * 17 * 128 + 19 * 128, device=dev)
w1, b1 = rnd(32, 16), rnd(32)
gate0 = torch.zeros(N * 1024, dtype=torch.int32, device=dev)
u = t_us(lambda: check(lib.clica_conv_k4s2_fwd_patches(P.data_ptr(), w1.data_ptr(), b1.data_ptr(), N, 16, 32, 32, 32, 1, 1, S2.data_ptr(), gate0.data_ptr(), st), "f1"))
print("stage1 fwd (patches) %.1f us  (writes %.0f MB: %.2f TB/s)" % (u, N * 1024 * 32 * 4 / 1e6, N * 1024 * 32 * 4 / u / 1e6))
for (C_, Co, ho) in stages[1:]:
    hs = ho + 1
    rows = N * hs * hs
    S = torch.rand(rows * 4 * C_ + (hs + 2) * 4 * C_, device=dev) - 0.5
    Wg, b = rnd(Co, 16 * C_), rnd(Co)
    nxt = torch.zeros(N * (ho // 2 + 1) ** 2 * 4 * Co + 64 * 4 * Co, device=dev)
    gbits = torch.zeros(rows * (Co // 32), dtype=torch.int32, device=dev)
    gf = 2.0 * rows * 16 * C_ * Co / 1e9
    u = t_us(lambda: check(lib.clica_conv_k4s2_fwd(S.data_ptr(), Wg.data_ptr(), b.data_ptr(), N, C_, Co, hs, hs, 1, 1, nxt.data_ptr(), gbits.data_ptr(), st), "f"))
    print("C=%d Cout=%d grid %d: fwd %.1f us = %.1f TFLOP/s (rows incl. non-outputs)" % (C_, Co, hs, u, gf / u * 1e3))
    front = (hs + 1) * Co
    dOs = torch.zeros(front + rows * Co, device=dev); dO = dOs[front:]; dO.normal_()
    Wd = rnd(4 * Co, 4 * C_)
    dgrid = 2 * ho if C_ == 32 and ho == 16 else 2 * ho + 1
    dprev = torch.zeros(N * dgrid * dgrid * C_, device=dev)
    pbits = torch.randint(-2 ** 31, 2 ** 31 - 1, (N * dgrid * dgrid * (C_ // 32),), dtype=torch.int32, device=dev)
    for gate in ("bits", "S", "none"):
        u = t_us(lambda: check(lib.clica_conv_k4s2_dgrad(dO.data_ptr(), Wd.data_ptr(), None if gate == "none" else S.data_ptr(), N, C_, Co, hs, hs, dprev.data_ptr(), dgrid, dgrid,
                                                         pbits.data_ptr() if gate == "bits" else None, st), "d"))
        print("   dgrad (gate: %s) %.1f us = %.1f TFLOP/s" % (gate, u, 2.0 * rows * 4 * Co * 4 * C_ / 1e9 / u * 1e3))
    nb = C.c_size_t(); check(lib.clica_conv_k4s2_wgrad_workspace_bytes(rows, Co, 16 * C_, C.byref(nb)), "ws")
    ws = workspace("probe", nb.value, dev); dW = torch.empty(Co, 16 * C_, device=dev); db = torch.empty(Co, device=dev)
    u = t_us(lambda: check(lib.clica_conv_k4s2_wgrad(dO.data_ptr(), S.data_ptr(), N, C_, Co, hs, hs, dW.data_ptr(), db.data_ptr(), 0, ws.data_ptr(), ws.numel(), st), "w"))
    print("   wgrad %.1f us = %.1f TFLOP/s" % (u, gf / u * 1e3))
