import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cl_ica_amd import ops
def dev(a): return torch.tensor(np.asarray(a, np.float32), device="cuda")
def run(dims, M, fill):
    rng = np.random.default_rng(1)
    L = len(dims) - 1
    Ws = [dev((rng.uniform(-1, 1, size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)) for i in range(L)]
    bs = [dev(rng.uniform(-0.5, 0.5, size=dims[i + 1]).astype(np.float32)) for i in range(L)]
    x = dev(rng.normal(size=(M, dims[0])).astype(np.float32))
    outs = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    masks = ops.mlp_signmask_alloc(M, L - 1, "cuda") + [None]
    kinds = [ops.mlp_wgrad_split_kind(dims[l + 1], dims[l]) for l in range(L)]
    act_pl = [ops.mlp_planes_alloc(M, dims[l + 1], True, "cuda") if (l + 1 < L and kinds[l + 1] == 0) else None for l in range(L)]
    dz_pl = [ops.mlp_planes_alloc(M, dims[l + 1], False, "cuda") if kinds[l] == 0 else None for l in range(L)]
    for t in act_pl + dz_pl:
        if t is not None: t.fill_(fill)
    packed, packed_t = ops.mlp_pack_split_both(Ws)
    ops.mlp_fwd_split(x, Ws, bs, outs, packed, 0.01, signmasks=masks, planes=act_pl)
    dy = dev(rng.normal(size=(M, dims[-1])).astype(np.float32))
    chain = list(range(L - 1, 0, -1))
    dz = [torch.empty(M, dims[l], device="cuda") for l in chain]
    ops.mlp_dgrad_chain_split(dy, [Ws[l] for l in chain], packed_t, dz, 0.01, masks_chain=[masks[l - 1] for l in chain], planes=[dz_pl[l - 1] for l in chain])
    torch.cuda.synchronize()
    # are the plane buffers fully overwritten? (count remaining fill bytes in the units the consumer reads)
    for nm, lst in (("act", act_pl), ("dz", dz_pl)):
        for l, t in enumerate(lst):
            if t is not None and fill != 0:
                used = t[: (M + 15) // 16 * (t.numel() // ((M + 47) // 48 * 3))]
                print(f"  {nm}[{l}] bytes still == fill in used groups: {int((used == fill).sum())} of {used.numel()}")
    dz_of = {l - 1: dz[j] for j, l in enumerate(chain)}; dz_of[L - 1] = dy
    dWs = [torch.full((dims[l + 1], dims[l]), 7.0, device="cuda") for l in range(L)]
    dbs = [torch.full((dims[l + 1],), 7.0, device="cuda") for l in range(L)]
    xs = [x] + outs[:-1]
    ops.mlp_wgrad_split(M, dz_pl, [act_pl[l - 1] if l > 0 else None for l in range(L)],
                        [dz_of[l] if kinds[l] == 1 else None for l in range(L)], [xs[l] if kinds[l] == 1 else None for l in range(L)], dWs, dbs)
    torch.cuda.synchronize()
    for l in range(L):
        w = dWs[l].cpu().numpy(); ref = dz_of[l].cpu().numpy().astype(np.float64).T @ xs[l].cpu().numpy().astype(np.float64)
        bad = ~np.isfinite(w)
        err = np.abs(np.where(bad, 0, w) - ref).max() / np.abs(ref).max()
        rows = np.where(bad.any(1))[0]; cols = np.where(bad.any(0))[0]
        print(f"  layer {l} kind {kinds[l]} shape {w.shape}: nonfinite {int(bad.sum())} rows {rows[:6]}..{rows[-3:] if len(rows) else ''} cols {cols[:6]}..{cols[-3:] if len(cols) else ''} err(finite part) {err:.2e}")
for fill in (0xFF, 0x00, 0x3F):
    for dims, M in (([4, 40, 200, 40, 4], 96), ([4, 40, 200, 40, 4], 1024)):
        print("fill", hex(fill), dims, M)
        run(dims, M, fill)


def decode_planes(buf, M, width, ones):
    """[groups][units][3][512] bf16 -> fp64 [groups*16][units*32] = hi + mid + lo"""
    units = (width + (1 if ones else 0) + 31) // 32
    groups = (M + 47) // 48 * 3
    raw = buf.cpu().numpy().view(np.uint16).reshape(groups, units, 3, 4, 2, 4, 16)      # [g][u][p][kq][s][kr][c]
    f = (raw.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    f = f.sum(2)                                                                         # [g][u][kq][s][kr][c]
    f = f.transpose(0, 2, 4, 1, 3, 5)                                                    # [g][kq][kr][u][s][c]
    return f.reshape(groups * 16, units * 32)


print("---- producer check")
rng = np.random.default_rng(1)
dims, M = [4, 40, 200, 40, 4], 96
L = len(dims) - 1
Ws = [dev((rng.uniform(-1, 1, size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)) for i in range(L)]
bs = [dev(rng.uniform(-0.5, 0.5, size=dims[i + 1]).astype(np.float32)) for i in range(L)]
x = dev(rng.normal(size=(M, dims[0])).astype(np.float32))
outs = [torch.empty(M, d, device="cuda") for d in dims[1:]]
masks = ops.mlp_signmask_alloc(M, L - 1, "cuda") + [None]
act_pl = [ops.mlp_planes_alloc(M, dims[l + 1], True, "cuda") for l in range(L)]
packed, packed_t = ops.mlp_pack_split_both(Ws)
ops.mlp_fwd_split(x, Ws, bs, outs, packed, 0.01, signmasks=masks, planes=act_pl)
torch.cuda.synchronize()
for l in range(L):
    dec = decode_planes(act_pl[l], M, dims[l + 1], True)
    o = outs[l].cpu().numpy().astype(np.float64)
    w = dims[l + 1]
    print(f"layer {l} width {w}: max |planes - fp32| over real rows/cols {np.abs(dec[:M, :w] - o).max():.3e}; ones column {dec[:M, w].min()} .. {dec[:M, w].max()};"
          f" padding cols max |.| {np.abs(dec[:M, w + 1:]).max() if dec.shape[1] > w + 1 else 0:.3e}; finite {np.isfinite(dec).all()}")
