#!/usr/bin/env python3
"""Effective shader clock while the training step (or one of its kernels) runs back to back: s_memtime delta / s_memrealtime delta.
usage: python tools/clock_probe.py   (builds tools/proto/clock_probe.hip with hipcc on the GPU box)"""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libclock_probe.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools/proto/clock_probe.hip")], check=True)
lib = ctypes.CDLL(so)
lib.clock_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]


def measure(fn, reps, label):
    a = torch.zeros(2, dtype=torch.int64, device="cuda"); b = torch.zeros(2, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(reps // 4): fn()
    torch.cuda.synchronize()
    lib.clock_stamp(a.data_ptr(), st)
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    lib.clock_stamp(b.data_ptr(), st)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    d = (b - a).cpu().tolist()
    wall_s = d[1] / 100e6
    print(f"{label:34s} {1e6 * wall_s / reps:9.1f} us/iter (host {1e6 * el / reps:7.1f})  shader clock {d[0] / wall_s / 1e9:.3f} GHz", flush=True)


import bench
args = bench.parse()
tr = bench.build_trainer(args, torch.device("cuda"), 1)
tr.capture()
measure(tr.step, 600, "whole step (graph replay)")
from cl_ica_amd import ops
R = 2 * tr.B
ws = [lin.weight for lin in tr.linears]; bs = [lin.bias for lin in tr.linears]
def g(fn):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    return gr.replay
measure(g(lambda: ops.mlp_fwd(tr.x, ws, bs, tr.acts, tr.slope, packed=tr.packed, signmasks=tr.signmasks)), 600, "mlp_fwd_k only")
L = len(tr.linears); order = list(range(L)); g_top = tr.dy
measure(g(lambda: ops.mlp_wgrad([g_top if l == L - 1 else tr.dz[l] for l in order], [tr.acts[l - 1] if l > 0 else tr.x for l in order],
                                [tr._gviews[id(tr.linears[l].weight)] for l in order], [tr._gviews[id(tr.linears[l].bias)] for l in order], ws=tr.group_ws)), 600, "grouped wgrad only")
z = torch.zeros_like(tr.x)
x0 = tr.x.clone(); tr.x.zero_()
measure(g(lambda: ops.mlp_fwd(tr.x, ws, bs, tr.acts, tr.slope, packed=tr.packed, signmasks=tr.signmasks)), 600, "mlp_fwd_k, zero input")
