#!/usr/bin/env python3
"""Effective shader clock while the training step (or one of its kernels) runs back to back: s_memtime delta / s_memrealtime delta.
(Round 4: stamps taken by separate one-thread launches can land on different CUs, whose core-clock counters do not share an origin
-- tools/proto/xcc_probe.hip; bench.py now uses a one-wave probe that stays on one CU, clica_clock_probe.  Kept for the rocm-smi
readout and the back-to-back kernel loops; treat its GHz column as indicative.)
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
rs = subprocess.run("rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v '^=' | head -30", shell=True, capture_output=True, text=True)
print(rs.stdout)
for native in (False, True):
    sys.argv = [sys.argv[0]] + (["--native-fp32"] if native else [])
    args = bench.parse()
    tr = bench.build_trainer(args, torch.device("cuda"), 1)
    tr.capture()
    tag = "native fp32" if native else "split-bf16"
    measure(tr.step, 800, f"[{tag}] whole step (graph replay)")
    g_top = tr.dy

    def g(fn):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        return gr.replay

    def fwd():
        tr._packed_current = True
        tr.forward()

    def chain():
        tr._packed_current = True
        tr.backward_chain(g_top)
    measure(g(fwd), 800, f"[{tag}] forward stack only")
    measure(g(chain), 800, f"[{tag}] backward chain only")
    measure(g(lambda: tr.weight_grads(g_top)), 800, f"[{tag}] weight gradients only")
    measure(g(tr.loss_forward_backward), 800, f"[{tag}] loss fwd + bwd only")
    measure(tr.step, 800, f"[{tag}] whole step again")
    del tr
    torch.cuda.empty_cache()
rs = subprocess.run("rocm-smi --showpower --showclocks 2>&1 | grep -v '^=' | head -30", shell=True, capture_output=True, text=True)
print(rs.stdout)
