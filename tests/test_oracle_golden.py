"""The oracle (oracle/np_oracle.py) against the golden vectors generated from the reference.

CPU only.  This is what pins the oracle: the reference itself has no tests.
Tolerances: the goldens are the reference's fp32 outputs, the oracle is fp64 -> a few fp32 ulps
of headroom relative to the largest element (observed <= 2.4e-7).
"""
import numpy as np
import pytest

from conftest import formula_weights, mlp_formula_params, rel_err
from oracle import np_oracle as O

TOL = 2e-6


def grad_scale(c):
    """Size of the un-cancelled positive-pair gradient term: (2 alpha / (B tau)) max |d|d|^p/dd|."""
    m = c["meta"]
    if "normalize" in m:      # dot-product InfoNCE: pull = (2 alpha / (B tau)) z2 (through d u / d z ~ 1 / |z| when normalised)
        z1 = np.asarray(c["in"]["z1"], np.float64); z2 = np.asarray(c["in"]["z2"], np.float64)
        mag = 1.0 / np.linalg.norm(z1, axis=1).min() if bool(m["normalize"]) else np.abs(z2).max()
        return 2 * float(m["alpha"]) / (z1.shape[0] * float(m["tau"])) * float(mag)
    if "p" not in m:
        return 0.0
    p = float(m["p"]); tau = float(m["tau"]); alpha = float(m["alpha"]) if "alpha" in m else 0.5
    d = np.abs(np.asarray(c["in"]["z1"], np.float64) - np.asarray(c["in"]["z2"], np.float64))
    if d.size == 0:
        return 0.0
    return 2 * alpha / (d.shape[0] * tau) * float((p * np.maximum(d, 1e-12) ** (p - 1)).max())


def _check_loss_case(c, out, tol=TOL, grads=("dz1", "dz2", "dz3")):
    # loss_i = 2(alpha*pos/tau + (1-alpha)*lse) cancels when the positive dominates the
    # softmax (lse ~ -pos/tau), so "relative" is taken w.r.t. the size of the two summands:
    # that is the fp32 rounding scale of the reference itself.
    B3 = c["in"]["z3"].shape[0] if "z3" in c["in"] else c["in"]["z1"].shape[0]
    comp = max(float(np.abs(out["loss_i"]).max()), float(np.abs(out["lse"]).max()) + np.log(B3 + 1.0), 1e-30)
    assert abs(float(out["loss_mean"]) - float(c["out"]["loss_mean"])) < tol * comp
    assert np.abs(out["loss_i"] - c["out"]["loss_i"]).max() < tol * comp
    if "pos_mean" in c["out"]:
        assert abs(float(out["pos_mean"]) - float(c["out"]["pos_mean"])) < tol * max(1.0, abs(float(c["out"]["pos_mean"])))
        assert abs(float(out["neg_mean"]) - float(c["out"]["neg_mean"])) < tol * max(1.0, abs(float(c["out"]["neg_mean"])))
    for g in grads:
        ref = c["out"][g]
        # gradients are a difference of two O(gscale) terms (alignment pull vs softmax push)
        scale = max(np.abs(ref).max(), grad_scale(c), 1e-30)
        # softmax weights exp(-neg/tau - lse) inherit the ABSOLUTE fp32 rounding of the exponent,
        # i.e. a relative error ~ eps * |lse| in the saturated (scale=3, p=3) cases
        sat = max(1.0, float(np.abs(out["lse"]).max()))
        assert np.abs(out[g] - ref).max() / scale < 5 * tol * sat, g


@pytest.mark.parametrize("name", ["g1_lp_loss.npz", "g2_rect.npz", "g3_misc.npz", "g19_wide_lp.npz", "g22_lp_loss_large.npz"])
def test_lp_loss_goldens(golden, name):
    G = golden(name)
    assert G.n_cases > 0
    for key, c in G.cases():
        m = c["meta"]
        out = O.lp_simclr_loss(c["in"]["z1"], c["in"]["z2"], c["in"]["z3"], p=float(m["p"]),
                               tau=float(m["tau"]), alpha=float(m["alpha"]),
                               compat=bool(m["compat"]), pow=bool(m["pow"]))
        _check_loss_case(c, out)


def test_lp_loss_roll_goldens(golden):
    G = golden("g1r_lp_roll.npz")
    for key, c in G.cases():
        m = c["meta"]
        z1, z2 = c["in"]["z1"], c["in"]["z2"]
        out = O.lp_simclr_loss(z1, z2, np.roll(z1, 1, 0), p=float(m["p"]), tau=float(m["tau"]),
                               alpha=float(m["alpha"]), compat=True, pow=True)
        out["dz1"] = out["dz1"] + np.roll(out["dz3"], -1, 0)
        _check_loss_case(c, out, grads=("dz1", "dz2"))


def test_analytic_kats():
    """All-zero embeddings: compat -> ln(B+1); default -> 0."""
    for B in (8, 512):
        z = np.zeros((B, 10))
        assert abs(O.lp_simclr_loss(z, z, z, 2, compat=True, grad=False)["loss_mean"] - np.log(B + 1)) < 1e-12
        assert abs(O.lp_simclr_loss(z, z, z, 2, compat=False, grad=False)["loss_mean"]) < 1e-12


def test_lp_loss_grad_finite_difference():
    """Independent check of the analytic backward incl. upstream grads on all four outputs."""
    rng = np.random.default_rng(0)
    z1 = rng.normal(size=(7, 3)); z2 = z1 + 0.1 * rng.normal(size=(7, 3)); z3 = rng.normal(size=(9, 3))
    gi = rng.normal(size=7)
    for p in (1, 2, 3, 1.5):
        for compat in (True, False):
            for pw in (True, False):
                kw = dict(p=p, tau=0.7, alpha=0.3, compat=compat, pow=pw)

                def scalar(a, b, c):
                    o = O.lp_simclr_loss(a, b, c, grad=False, **kw)
                    return 1.3 * o["loss_mean"] + (gi * o["loss_i"]).sum() + 0.4 * o["pos_mean"] - 0.2 * o["neg_mean"]
                out = O.lp_simclr_loss(z1, z2, z3, g_mean=1.3, g_item=gi, g_pos=0.4, g_neg=-0.2, **kw)
                for name, arr, idx in (("dz1", z1, 0), ("dz2", z2, 1), ("dz3", z3, 2)):
                    num = np.zeros_like(arr)
                    for i in np.ndindex(arr.shape):
                        args = [z1.copy(), z2.copy(), z3.copy()]
                        args[idx][i] += 1e-6; up = scalar(*args)
                        args[idx][i] -= 2e-6; dn = scalar(*args)
                        num[i] = (up - dn) / 2e-6
                    assert np.abs(num - out[name]).max() < 1e-6, (p, compat, pw, name)


@pytest.mark.parametrize("name", ["g5_simclr.npz", "g19_wide_simclr.npz"])
def test_simclr_goldens(golden, name):
    G = golden(name)
    for key, c in G.cases():
        m = c["meta"]
        out = O.simclr_loss(c["in"]["z1"], c["in"]["z2"], c["in"]["z3"], normalize=bool(m["normalize"]),
                            tau=float(m["tau"]), alpha=float(m["alpha"]))
        _check_loss_case(c, out)


def test_mlp_goldens(golden):
    G = golden("g6_mlp.npz")
    for key, c in G.cases():
        n = int(c["meta"]["n"]); head = str(c["meta"]["head"]); hidden = [int(h) for h in c["meta"]["hidden"]]
        head = None if head == "None" else head
        Ws, bs, hp = mlp_formula_params(n, hidden, head)
        P = O.MLPParams(Ws, bs, head, hp)
        y, cache = O.mlp_forward(P, c["in"]["x"])
        assert rel_err(y, c["out"]["y"]) < 5e-6, key
        gr = O.mlp_backward(P, cache, c["in"]["gy"])
        assert rel_err(gr["dx"], c["out"]["dx"]) < 2e-5, key
        for l in range(len(Ws)):
            for kind, got in (("weight", gr["dW"][l]), ("bias", gr["db"][l])):
                nm = f"{2 * l}.{kind}"
                if f"grad/{nm}" in c["out"]:
                    assert rel_err(got, c["out"][f"grad/{nm}"]) < 2e-5, (key, nm)
                else:
                    sub = np.ascontiguousarray(got.reshape(-1)[::97])
                    assert rel_err(sub, c["out"][f"gradsub/{nm}"]) < 2e-5, (key, nm)
                    s = c["out"][f"gradsum/{nm}"]
                    assert abs(got.sum() - s[0]) < 2e-5 * max(1.0, np.abs(got).sum())
        last = 2 * len(Ws) - 1
        if head == "learnable_sphere":
            assert rel_err(gr["dhead"], c["out"][f"grad/{last}.r"]) < 2e-5
        if head == "learnable_box":
            assert rel_err(gr["dhead"], c["out"][f"grad/{last}.max_abs_bound"]) < 2e-5
        keys = [str(k) for k in c["meta"]["state_keys"]]
        assert keys[:2] == ["0.weight", "0.bias"]


def test_mixing_golden(golden):
    z = golden("g8_mixing.npz").z
    y = O.mixing_forward([z["W0"], z["W1"], z["W2"]], z["x"])
    assert rel_err(y, z["y"]) < 2e-6


def test_trainstep_goldens(golden):
    G = golden("g7_trainstep.npz")
    for key, c in G.cases():
        p = int(c["meta"]["p"]); head = str(c["meta"]["head"]); head = None if head == "None" else head
        hidden = [int(h) for h in c["meta"]["hidden"]]; n = 4
        Ws, bs, hp = mlp_formula_params(n, hidden, head)
        P = O.MLPParams(Ws, bs, head, hp)
        nparam = 2 * len(Ws) + (1 if head in ("learnable_sphere", "learnable_box") else 0)
        shapes = []
        for l in range(len(Ws)):
            shapes += [Ws[l].shape, bs[l].shape]
        if nparam > 2 * len(Ws):
            shapes.append(hp.shape)
        st = dict(step=0, m=[np.zeros(s) for s in shapes], v=[np.zeros(s) for s in shapes])
        gWs = [c["in"][f"g{i}"] for i in range(3)]
        for s in range(5):
            loss, pm, nm = O.train_step(P, gWs, c["in"][f"z1_{s}"], c["in"][f"z2_{s}"], st, p=p,
                                        lr=float(c["meta"]["lr"]))
            assert abs(loss - c["out"]["loss"][s]) < 1e-5 * abs(c["out"]["loss"][s]), (key, s)
            assert abs(pm - c["out"]["pos"][s]) < 1e-5 * max(1.0, abs(c["out"]["pos"][s]))
            assert abs(nm - c["out"]["neg"][s]) < 1e-5 * max(1.0, abs(c["out"]["neg"][s]))
        for l in range(len(Ws)):
            for kind, got in (("weight", P.W[l]), ("bias", P.b[l])):
                ref = c["out"][f"param5/{2 * l}.{kind}"]
                got = got if ref.size == got.size else np.ascontiguousarray(got.reshape(-1)[::7])
                # Adam's first steps move every weight by ~lr regardless of gradient size, so
                # sign flips of ~0 gradients make a few elements differ by O(lr): compare in bulk
                diff = np.abs(got.reshape(-1) - ref.reshape(-1))
                if l == len(Ws) - 1 and kind == "bias" and head is None:
                    # Lp distances are translation invariant: the exact gradient of the last bias
                    # is 0, the computed one is rounding noise, and Adam turns noise into +-lr
                    # steps.  Only a random-walk bound is meaningful here.
                    assert diff.max() <= 5 * float(c["meta"]["lr"]) * 1.01
                    continue
                assert np.median(diff) < 1e-6, (key, l, kind)
                assert (diff > 2e-4).mean() < 0.01, (key, l, kind, float((diff > 2e-4).mean()))


def test_supervised_phase_goldens(golden):
    """G24 (tests/golden/gen_goldens_r3.py): the reference's SUPERVISED train_step (F.mse_loss(z1_rec, z1), main_mlp.py:274-276)
    pins oracle.supervised_train_step: per-step loss, step-0 gradients, and the parameters after five Adam updates on the
    elements whose gradient stayed above 1 % of the tensor's largest (g24's own gmin / gmax record)."""
    G = golden("g24_supervised.npz")
    for key, c in G.cases():
        n = int(c["meta"]["n"]); head = str(c["meta"]["head"]); head = None if head == "None" else head
        hidden = [int(h) for h in c["meta"]["hidden"]]; lr = float(c["meta"]["lr"]); stride = int(c["meta"]["stride"])
        Ws, bs, hp = mlp_formula_params(n, hidden, head)
        P = O.MLPParams(Ws, bs, head, hp)
        shapes = []
        for l in range(len(Ws)):
            shapes += [Ws[l].shape, bs[l].shape]
        if head in ("learnable_sphere", "learnable_box"):
            shapes.append(hp.shape)
        st = dict(step=0, m=[np.zeros(s) for s in shapes], v=[np.zeros(s) for s in shapes])
        gWs = [c["in"][f"g{i}"] for i in range(3)]
        names = [f"{2 * l}.{kind}" for l in range(len(Ws)) for kind in ("weight", "bias")]
        if len(shapes) > len(names):
            names.append(f"{2 * len(Ws) - 1}.max_abs_bound" if head == "learnable_box" else f"{2 * len(Ws) - 1}.r")
        for s in range(int(c["meta"]["steps"])):
            loss, grads = O.supervised_train_step(P, gWs, c["in"][f"z1_{s}"], st, lr=lr, return_grads=True)
            assert abs(loss - c["out"]["loss"][s]) < 1e-5 * abs(c["out"]["loss"][s]), (key, s, loss, c["out"]["loss"][s])
            if s == 0:
                for name, g in zip(names, grads):
                    ref = c["out"][f"grad0/{name}"]
                    got = g if ref.size == g.size else np.ascontiguousarray(np.asarray(g).reshape(-1)[::stride])
                    assert rel_err(got.reshape(-1), ref.reshape(-1)) < 2e-5, (key, name)
        finals = []
        for l in range(len(Ws)):
            finals += [P.W[l], P.b[l]]
        if len(shapes) > 2 * len(Ws):
            finals.append(P.head_param)
        n_strict = 0
        for name, got in zip(names, finals):
            ref = c["out"][f"param5/{name}"]
            got = got if ref.size == got.size else np.ascontiguousarray(got.reshape(-1)[::stride])
            strict = c["out"][f"gmin/{name}"].reshape(-1) > 0.01 * float(c["out"][f"gmax/{name}"])
            n_strict += int(strict.sum())
            if strict.any():
                err = np.abs(got.reshape(-1) - ref.reshape(-1))[strict].max() / max(np.abs(ref).max(), 1e-30)
                assert err < 6e-5, (key, name, err)
        assert n_strict > 1000


def test_adam_mask_file_matches_the_trajectory_goldens(golden):
    """g23_adam_masks.npz holds one gmin array per stored final-parameter array of G7 / G13 / G14 (solver cases) / G15, same shape."""
    z = golden("g23_adam_masks.npz").z
    for tag, fname, prefix in (("g7", "g7_trainstep.npz", "param5"), ("g13", "g13_wide_trainstep.npz", "paramN"),
                               ("g14", "g14_kitti.npz", "param3"), ("g15", "g15_3dident.npz", "param3")):
        gz = golden(fname).z
        keys = [k for k in gz.files if f"/out/{prefix}/" in k]
        assert keys
        for k in keys:
            case, name = k.split("/")[0], k.split(f"/out/{prefix}/")[1]
            m = z[f"{tag}/{case}/gmin/{name}"]
            assert m.shape == gz[k].shape and (m >= 0).all() and float(z[f"{tag}/{case}/gmax/{name}"]) >= float(m.max()) * 0.999999


def test_formula_in_sync(golden):
    c = golden("g6_mlp.npz").case("c000")
    assert np.array_equal(c["in"]["x"], np.asarray(formula_weights((48, 4), 99) * np.sqrt(4) * 1.5, np.float32))


def test_torch_port_goldens(golden):
    """The cpu_baseline port (oracle/torch_port.py) reproduces the reference's outputs."""
    import torch
    from oracle import torch_port as T
    G = golden("g1_lp_loss.npz")
    n = 0
    for key, c in G.cases():
        m = c["meta"]
        if c["in"]["z1"].shape[0] > 64:
            continue
        a, b, cc = (torch.tensor(c["in"][k], requires_grad=True) for k in ("z1", "z2", "z3"))
        mean, per, (pm, nm) = T.lp_simclr_loss(a, b, cc, p=int(m["p"]), tau=float(m["tau"]), alpha=float(m["alpha"]),
                                               compat=bool(m["compat"]), pow=bool(m["pow"]))
        mean.backward()
        assert np.array_equal(per.detach().numpy(), c["out"]["loss_i"]) or rel_err(per.detach().numpy(), c["out"]["loss_i"]) < 1e-6
        assert rel_err(a.grad.numpy(), c["out"]["dz1"]) < 1e-5
        n += 1
    assert n > 50


@pytest.mark.parametrize("gname", ["g12_align_uniform.npz", "g18_align_uniform_frac.npz"])
def test_uniformity_alignment_oracle_vs_golden(golden, gname):
    """G12 / G18 (fractional exponents): UniformityLoss / AlignmentLoss (losses.py:205-241)."""
    G = golden(gname)
    for i in range(G.n_cases):
        u = G.case(f"u{i:03d}"); p = float(u["meta"]["p"])
        out = O.uniformity_loss(u["in"]["z1"], u["in"]["z3"], p)
        comp = max(float(np.abs(out["loss_i"]).max()) + np.log(u["in"]["z1"].shape[0]), 1.0)
        assert abs(float(out["loss_mean"]) - float(u["out"]["loss_mean"])) < TOL * comp
        assert np.abs(out["loss_i"] - u["out"]["loss_i"]).max() < TOL * comp
        for g in ("dz1", "dz3"):
            # rows of softmax weights sum to 1/B3: that is the un-cancelled size of a gradient entry
            scale = max(np.abs(u["out"][g]).max(), float(np.abs(O._dpowabs(u["in"]["z1"][None] - u["in"]["z3"][:, None], p)).max()) / u["in"]["z3"].shape[0])
            assert np.abs(out[g] - u["out"][g]).max() / scale < 5 * TOL * comp, (i, g)
        a = G.case(f"a{i:03d}")
        out = O.alignment_loss(a["in"]["z1"], a["in"]["z2"], p)
        assert abs(float(out["loss_mean"]) - float(a["out"]["loss_mean"])) < TOL * max(1.0, abs(float(a["out"]["loss_mean"])))
        assert rel_err(out["loss_i"], a["out"]["loss_i"]) < TOL
        for g in ("dz1", "dz2"):
            assert rel_err(out[g], a["out"][g]) < 5 * TOL, (i, g)


def test_flat_l2_search_known_answers():
    """oracle.flat_l2_search on a lattice where the answers are known by construction (faiss.IndexFlatL2 semantics:
    ascending squared distances, positions in add order; ties to the lower row)."""
    g = np.stack(np.meshgrid(np.arange(5), np.arange(4), np.arange(3), indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    q = g[[7, 31, 59]] + np.array([[0.1, -0.2, 0.05], [0.3, 0.1, -0.1], [-0.2, -0.3, 0.1]])
    D, I = O.flat_l2_search(g, q, 2)
    assert I[:, 0].tolist() == [7, 31, 59]
    assert np.allclose(D[:, 0], [0.1 ** 2 + 0.2 ** 2 + 0.05 ** 2, 0.3 ** 2 + 0.1 ** 2 + 0.1 ** 2, 0.2 ** 2 + 0.3 ** 2 + 0.1 ** 2])
    assert np.all(D[:, 1] >= D[:, 0])
    # a query in the middle of a lattice edge: both ends at distance 0.25, the lower row first
    D, I = O.flat_l2_search(g, np.array([[1.0, 1.0, 0.5]]), 2)
    a, b = int(np.where((g == [1, 1, 0]).all(1))[0][0]), int(np.where((g == [1, 1, 1]).all(1))[0][0])
    assert I[0].tolist() == [min(a, b), max(a, b)] and np.allclose(D[0], 0.25)
    # the dataset rule (threedident_dataset.py:108-113): z~ snapping onto z's grid point takes its second neighbour
    iz, izt = O.threedident_snap(g, g[[7]] + 0.01, g[[7]] + 0.02)
    assert iz[0] == 7 and izt[0] != 7


def test_kitti_pair_pipeline_known_answers():
    """oracle.kitti_getitem / kitti_collate against HAND-COMPUTED answers of the reference's published semantics (the module itself needs
    torchvision / matplotlib, absent here, so this half of row N4 is pinned to the text of kitti_masks/dataset.py, not to its execution):
      * dataset.py:90-95  a flat index addresses sequence s = searchsorted(cumlens, index, side="right") at frame index - cumlens[s - 1]
                          (so index == cumlens[s - 1] is frame 0 of sequence s, NOT the last frame of s - 1);
      * dataset.py:97-98  the partner frame is min(start + t, len - 1): clipped to the sequence's last frame, never into the next one;
      * dataset.py:100-131 masks become uint8 x 255, get a channel axis and come back as float32 in {0, 1};
      * dataset.py:134-142 custom_collate interleaves: inputs = [first_0, second_0, first_1, second_1, ...], labels likewise."""
    # three sequences of 3, 2 and 4 frames of 2 x 2 masks; frame f of sequence s is filled with the bit pattern of (10 s + f)
    lens = [3, 2, 4]
    data = [[np.array([[(10 * s + f) & 1, ((10 * s + f) >> 1) & 1], [((10 * s + f) >> 2) & 1, ((10 * s + f) >> 3) & 1]], bool) for f in range(n)]
            for s, n in enumerate(lens)]
    lat = [[np.array([s, f, 10 * s + f], np.float32) for f in range(n)] for s, n in enumerate(lens)]
    cum = np.cumsum(lens)                                                   # [3, 5, 9]
    # index 0: sequence 0 frame 0, t = 1 -> frame 1
    a, b, la, lb = O.kitti_getitem(data, lat, cum, 0, 1)
    assert a.shape == (1, 2, 2) and a.dtype == np.float32 and la.tolist() == [0, 0, 0] and lb.tolist() == [0, 1, 1]
    assert a.ravel().tolist() == [0, 0, 0, 0] and b.ravel().tolist() == [1, 0, 0, 0]
    # index 2: last frame of sequence 0, t = 5 -> clipped to itself (dataset.py:98), never frame 0 of sequence 1
    a, b, la, lb = O.kitti_getitem(data, lat, cum, 2, 5)
    assert la.tolist() == lb.tolist() == [0, 2, 2] and np.array_equal(a, b)
    # index 3 == cumlens[0]: frame 0 of sequence 1 (side="right", dataset.py:90), t = 1 -> frame 1 of sequence 1
    a, b, la, lb = O.kitti_getitem(data, lat, cum, 3, 1)
    assert la.tolist() == [1, 0, 10] and lb.tolist() == [1, 1, 11]
    assert a.ravel().tolist() == [0, 1, 0, 1] and b.ravel().tolist() == [1, 1, 0, 1]          # 10 = 0b1010, 11 = 0b1011 (bit 0 first)
    # index 6: sequence 2 frame 1, t = 2 -> frame 3 (the last one, exactly)
    a, b, la, lb = O.kitti_getitem(data, lat, cum, 6, 2)
    assert la.tolist() == [2, 1, 21] and lb.tolist() == [2, 3, 23]
    # collate (dataset.py:134-142): interleaved pairs, labels in the same order
    s0, s1 = O.kitti_getitem(data, lat, cum, 0, 1), O.kitti_getitem(data, lat, cum, 6, 2)
    x, y = O.kitti_collate([s0, s1])
    assert x.shape == (4, 1, 2, 2) and y.shape == (4, 3)
    assert y[:, 2].tolist() == [0, 1, 21, 23]
    assert np.array_equal(x[0], s0[0]) and np.array_equal(x[1], s0[1]) and np.array_equal(x[2], s1[0]) and np.array_equal(x[3], s1[1])
    # the solver then splits the batch back with mu[::2] / mu[1::2] (kitti_masks/solver.py:64-65): anchors 0, 21 and partners 1, 23
    assert y[::2, 2].tolist() == [0, 21] and y[1::2, 2].tolist() == [1, 23]


def test_flat_l2_search_published_faiss_semantics():
    """faiss.IndexFlatL2 (threedident_dataset.py:69, 104-105; faiss absent: pinned to its published definition) returns the EXACT k smallest
    SQUARED Euclidean distances in ascending order, ids in add() order, and among equal distances the smaller id first -- hand-computed on
    a five-row table with two exact ties."""
    table = np.array([[0.0, 0.0], [3.0, 4.0], [-3.0, 4.0], [6.0, 8.0], [0.0, 5.0]])
    D, I = O.flat_l2_search(table, np.array([[0.0, 0.0], [0.0, 4.0]]), 4)
    assert I[0].tolist() == [0, 1, 2, 4] and D[0].tolist() == [0.0, 25.0, 25.0, 25.0]        # three rows at squared distance 25: ids ascending
    assert I[1].tolist() == [4, 1, 2, 0] and D[1].tolist() == [1.0, 9.0, 9.0, 16.0]           # SQUARED distances (not 1, 3, 3, 4)
