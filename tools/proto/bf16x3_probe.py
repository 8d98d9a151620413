#!/usr/bin/env python3
"""Prototype probe: one 500x500 Linear over 12288 rows as six bf16 MFMA products of exact 3-way bf16 splits
(tools/proto/bf16x3_layer.hip) vs the product's fp32-MFMA kernels: accuracy against fp64 and time."""
import ctypes as C, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
so = os.path.join(here, "libbf16x3.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so,
                    os.path.join(here, "bf16x3_layer.hip")], check=True)
lib = C.CDLL(so)
from cl_ica_amd import ops
from tools.mlp_bench import replay_time

def split3(w):
    b = w.contiguous().view(torch.int32)
    hb = b & -65536; h = hb.view(torch.float32); r1 = w - h
    mb = r1.view(torch.int32) & -65536; m = mb.view(torch.float32); r2 = r1 - m
    lb = r2.view(torch.int32) & -65536
    return [(x >> 16).to(torch.int16) for x in (hb, mb, lb)]

def pack(W):
    N, K = W.shape
    ncb, kit = (N + 15) // 16, (K + 31) // 32
    Wp = torch.zeros(ncb * 16, kit * 32, device=W.device); Wp[:N, :K] = W
    out = []
    for piece in split3(Wp):
        t = piece.view(ncb, 16, kit, 4, 8).permute(0, 2, 3, 1, 4).contiguous()      # [cb][ki][kg][i15][8]
        out.append(t.reshape(-1))
    return torch.cat(out).contiguous()

M, K, N = 12288, 500, 500
torch.manual_seed(0)
X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
ref = (X.double() @ W.double().T)
Wp = pack(W)
Y = torch.zeros(M, N, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for products in (6, 3, 1):
    Y.zero_()
    rc = lib.bf16x3_layer(C.c_void_p(X.data_ptr()), C.c_int64(K), C.c_int64(M), K, N, C.c_void_p(Wp.data_ptr()), C.c_void_p(Y.data_ptr()),
                          C.c_int64(N), products, 1, C.c_void_p(st))
    torch.cuda.synchronize(); assert rc == 0
    err = ((Y.double() - ref).abs().max() / ref.abs().max()).item()
    t = replay_time(lambda: lib.bf16x3_layer(C.c_void_p(X.data_ptr()), C.c_int64(K), C.c_int64(M), K, N, C.c_void_p(Wp.data_ptr()),
                                             C.c_void_p(Y.data_ptr()), C.c_int64(N), products, 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    print(f"bf16 split, {products} products: max err/max|ref| {err:.2e}   {t:7.1f} us  ({2.0*M*N*K/t/1e6:6.1f} fp32-equivalent TFLOP/s)")
for products in (6, 1):
    ts = []
    for rep in (1, 2, 4, 8):
        ts.append(replay_time(lambda: lib.bf16x3_layer(C.c_void_p(X.data_ptr()), C.c_int64(K), C.c_int64(M), K, N, C.c_void_p(Wp.data_ptr()),
                                                       C.c_void_p(Y.data_ptr()), C.c_int64(N), products, rep,
                                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))))
    print(f"{products} products, layer repeated 1/2/4/8 times: " + " ".join(f"{t:7.1f}" for t in ts) +
          f" us   marginal {(ts[3]-ts[0])/7:6.1f} us/layer")
b = torch.zeros(N, device="cuda")
y32 = ops.linear_fwd(X, W, b, False)
err32 = ((y32.double() - ref).abs().max() / ref.abs().max()).item()
t32 = replay_time(lambda: ops.linear_fwd(X, W, b, False, out=y32))
outs = [torch.empty(M, N, device="cuda")]; packed = ops.mlp_pack_weights([W])
tf = replay_time(lambda: ops.mlp_fwd(X, [W], [b], outs, 0.01, packed=packed))
print(f"fp32 MFMA per-layer GEMM   : max err/max|ref| {err32:.2e}   {t32:7.1f} us  ({2.0*M*N*K/t32/1e6:6.1f} TFLOP/s)")
print(f"fp32 MFMA fused (1 layer)  :                              {tf:7.1f} us  ({2.0*M*N*K/tf/1e6:6.1f} TFLOP/s)")
