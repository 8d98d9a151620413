#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (read-only at /root/reference).

Run only in the build container (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_goldens.py

Every fixture stores the INPUTS together with the reference's OUTPUTS, so nothing at
test time depends on torch's RNG streams.  Groups follow SURVEY.md section 8(c):

  g1_lp_loss.npz      LpSimCLRLoss grid (p, tau, compat, alpha, shape, scale), z3 leaf
  g1r_lp_roll.npz     LpSimCLRLoss with z3 = roll(z1) inside the graph (main_mlp.py:272)
  g2_rect.npz         rectangular B != B3
  g3_misc.npz         pow=False, p=0.5 (eps branch), analytic zeros KATs
  g5_simclr.npz       SimCLRLoss (dot-product InfoNCE)
  g6_mlp.npz          get_mlp forward/backward with formula-initialised weights + heads
  g7_trainstep.npz    5 injected train steps (loss floats + final weights after Adam)
  g8_mixing.npz       construct_invertible_mlp KAT + forward
  g9_samplers.npz     sampler statistics from 1e5 reference draws
  g10_strided.npz     strided / sliced input views
  g11_metrics.npz     R^2 / MCC evaluation metrics (disentanglement_utils.py)
  g12_align_uniform.npz  UniformityLoss / AlignmentLoss (losses.py:205-241)
"""
import os
import sys
import io
import contextlib

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import torch

import losses as ref_losses  # noqa: E402  (reference module)
import encoders as ref_encoders  # noqa: E402
import layers as ref_layers  # noqa: E402
import invertible_network_utils as ref_inu  # noqa: E402
import spaces as ref_spaces  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def t2n(t):
    return t.detach().cpu().numpy().astype(np.float32)


def run_loss(loss_obj, z1, z2, z3, upstream=None):
    """Call the reference loss on leaf copies; return outputs + grads w.r.t. each input."""
    a = torch.tensor(z1, requires_grad=True)
    b = torch.tensor(z2, requires_grad=True)
    c = torch.tensor(z3, requires_grad=True)
    mean, per_item, (pos_m, neg_m) = loss_obj(None, None, None, a, b, c)
    mean.backward()
    return dict(
        loss_mean=t2n(mean), loss_i=t2n(per_item), pos_mean=t2n(pos_m), neg_mean=t2n(neg_m),
        dz1=t2n(a.grad), dz2=t2n(b.grad), dz3=t2n(c.grad),
    )


def rand_inputs(seed, B, B3, n, scale):
    g = torch.Generator().manual_seed(seed)
    z1 = torch.randn(B, n, generator=g) * scale
    z2 = z1 + 0.05 * scale * torch.randn(B, n, generator=g)
    z3 = torch.randn(B3, n, generator=g) * scale
    return t2n(z1), t2n(z2), t2n(z3)


def save(name, store):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **store)
    print(f"{name}: {len(store)} arrays, {os.path.getsize(path)/1024:.1f} KiB")


def put(store, key, inputs, outputs, **meta):
    for k, v in inputs.items():
        store[f"{key}/in/{k}"] = v
    for k, v in outputs.items():
        store[f"{key}/out/{k}"] = v
    for k, v in meta.items():
        store[f"{key}/meta/{k}"] = np.asarray(v)


# ----------------------------------------------------------------------------- G1
def g1():
    store = {}
    idx = 0
    for (B, n) in [(8, 3), (64, 10)]:
        for p in (1, 2, 3):
            for tau in (1.0, 0.5):
                for compat in (True, False):
                    for alpha in (0.5, 0.3):
                        for scale in (0.1, 1.0, 3.0):
                            if B == 64 and (tau == 1.0) != (alpha == 0.5):
                                continue  # (64,10): pair (tau, alpha) as (1.0,0.5)/(0.5,0.3) only
                            z1, z2, z3 = rand_inputs(1000 + idx, B, B, n, scale)
                            L = ref_losses.LpSimCLRLoss(p=p, tau=tau, alpha=alpha,
                                                        simclr_compatibility_mode=compat)
                            out = run_loss(L, z1, z2, z3)
                            put(store, f"c{idx:03d}", dict(z1=z1, z2=z2, z3=z3), out,
                                p=p, tau=tau, compat=compat, alpha=alpha, pow=True, scale=scale)
                            idx += 1
    # the larger shapes on a reduced grid
    for (B, n) in [(512, 10), (256, 40)]:
        for p in (1, 2, 3):
            for (tau, compat, alpha, scale) in [(1.0, True, 0.5, 1.0), (0.5, False, 0.3, 0.1)]:
                if (B == 256 and p == 3) or (scale == 0.1 and p != 2):
                    continue
                z1, z2, z3 = rand_inputs(1000 + idx, B, B, n, scale)
                L = ref_losses.LpSimCLRLoss(p=p, tau=tau, alpha=alpha,
                                            simclr_compatibility_mode=compat)
                out = run_loss(L, z1, z2, z3)
                put(store, f"c{idx:03d}", dict(z1=z1, z2=z2, z3=z3), out,
                    p=p, tau=tau, compat=compat, alpha=alpha, pow=True, scale=scale)
                idx += 1
    store["n_cases"] = np.asarray(idx)
    save("g1_lp_loss.npz", store)


def g1_roll():
    """z3 = roll(z1_rec, 1, 0) inside the autograd graph (main_mlp.py:272): combined dz1."""
    store = {}
    idx = 0
    for (B, n) in [(8, 3), (64, 10), (512, 10), (256, 40)]:
        for p in (1, 2, 3):
            for scale in (0.1, 1.0, 3.0):
                if B >= 256 and scale != 1.0:
                    continue
                z1, z2, _ = rand_inputs(2000 + idx, B, B, n, scale)
                a = torch.tensor(z1, requires_grad=True)
                b = torch.tensor(z2, requires_grad=True)
                L = ref_losses.LpSimCLRLoss(p=p, tau=1.0, simclr_compatibility_mode=True)
                mean, per_item, (pm, nm) = L(None, None, None, a, b, torch.roll(a, 1, 0))
                mean.backward()
                put(store, f"c{idx:03d}", dict(z1=z1, z2=z2),
                    dict(loss_mean=t2n(mean), loss_i=t2n(per_item), pos_mean=t2n(pm),
                         neg_mean=t2n(nm), dz1=t2n(a.grad), dz2=t2n(b.grad)),
                    p=p, tau=1.0, compat=True, alpha=0.5, pow=True, scale=scale)
                idx += 1
    store["n_cases"] = np.asarray(idx)
    save("g1r_lp_roll.npz", store)


# ----------------------------------------------------------------------------- G2/G3/G4
def g2():
    store = {}
    idx = 0
    for p in (1, 2, 3):
        for compat in (True, False):
            z1, z2, z3 = rand_inputs(3000 + idx, 16, 64, 5, 1.0)
            L = ref_losses.LpSimCLRLoss(p=p, tau=0.7, simclr_compatibility_mode=compat)
            put(store, f"c{idx:03d}", dict(z1=z1, z2=z2, z3=z3), run_loss(L, z1, z2, z3),
                p=p, tau=0.7, compat=compat, alpha=0.5, pow=True)
            idx += 1
    store["n_cases"] = np.asarray(idx)
    save("g2_rect.npz", store)


def g3():
    store = {}
    idx = 0
    # pow=False (plain Lp norm): distinct inputs so the norm derivative is regular
    for p in (1, 2, 3):
        for compat in (True, False):
            z1, z2, z3 = rand_inputs(4000 + idx, 32, 32, 6, 1.0)
            L = ref_losses.LpSimCLRLoss(p=p, tau=1.0, simclr_compatibility_mode=compat, pow=False)
            put(store, f"c{idx:03d}", dict(z1=z1, z2=z2, z3=z3), run_loss(L, z1, z2, z3),
                p=p, tau=1.0, compat=compat, alpha=0.5, pow=False)
            idx += 1
    # fractional p < 1: eps branch with transposed orientation (losses.py:433-442)
    for pw in (True, False):
        for compat in (True, False):
            z1, z2, z3 = rand_inputs(4100 + idx, 6, 6, 4, 1.0)
            L = ref_losses.LpSimCLRLoss(p=0.5, tau=1.0, simclr_compatibility_mode=compat, pow=pw)
            put(store, f"c{idx:03d}", dict(z1=z1, z2=z2, z3=z3), run_loss(L, z1, z2, z3),
                p=0.5, tau=1.0, compat=compat, alpha=0.5, pow=pw)
            idx += 1
    # analytic KATs: all-zero embeddings => compat ln(B+1), default 0.0
    for B in (8, 512):
        for compat in (True, False):
            z = np.zeros((B, 10), np.float32)
            L = ref_losses.LpSimCLRLoss(p=2, tau=1.0, simclr_compatibility_mode=compat)
            put(store, f"c{idx:03d}", dict(z1=z, z2=z, z3=z), run_loss(L, z, z, z),
                p=2, tau=1.0, compat=compat, alpha=0.5, pow=True)
            idx += 1
    # z3 = roll(z1) as independent leaf with exact-zero distances, pow=True (p=1 sign(0)=0)
    for p in (1, 2, 3):
        z1, z2, _ = rand_inputs(4200 + idx, 16, 16, 4, 1.0)
        z3 = np.roll(z1, 1, 0).copy()
        L = ref_losses.LpSimCLRLoss(p=p, tau=1.0, simclr_compatibility_mode=True)
        put(store, f"c{idx:03d}", dict(z1=z1, z2=z2, z3=z3), run_loss(L, z1, z2, z3),
            p=p, tau=1.0, compat=True, alpha=0.5, pow=True)
        idx += 1
    store["n_cases"] = np.asarray(idx)
    save("g3_misc.npz", store)


# ----------------------------------------------------------------------------- G5
def g5():
    store = {}
    idx = 0
    for (B, B3, n) in [(8, 8, 3), (64, 64, 10), (16, 48, 5)]:
        for normalize in (True, False):
            for tau in (1.0, 0.5):
                for alpha in (0.5, 0.3):
                    z1, z2, z3 = rand_inputs(5000 + idx, B, B3, n, 1.0)
                    L = ref_losses.SimCLRLoss(normalize=normalize, tau=tau, alpha=alpha)
                    put(store, f"c{idx:03d}", dict(z1=z1, z2=z2, z3=z3), run_loss(L, z1, z2, z3),
                        normalize=normalize, tau=tau, alpha=alpha)
                    idx += 1
    store["n_cases"] = np.asarray(idx)
    save("g5_simclr.npz", store)


# ----------------------------------------------------------------------------- G6
def formula_weights(shape, salt):
    """Deterministic, RNG-free weights: smooth pseudo-random pattern scaled like nn.Linear."""
    fan_in = shape[-1] if len(shape) > 1 else shape[0]
    idx = np.arange(int(np.prod(shape)), dtype=np.float64).reshape(shape)
    w = np.sin(idx * 12.9898 + salt * 78.233) * 43758.5453
    w = w - np.floor(w)  # in [0,1)
    return ((2.0 * w - 1.0) / np.sqrt(fan_in)).astype(np.float32)


def f32(a):
    return np.asarray(a, dtype=np.float32)


def fill_formula(module):
    k = 0
    for name, prm in module.named_parameters():
        if name.endswith("weight") or name.endswith("bias"):
            prm.data = torch.tensor(formula_weights(tuple(prm.shape), k + 1))
            k += 1


def subsample(a, step=97):
    return np.ascontiguousarray(a.reshape(-1)[::step])


def g6():
    """get_mlp fwd/bwd.  Parameters are NOT stored: tests rebuild them with formula_weights
    (same enumeration order as fill_formula), so only outputs/gradients are kept."""
    store = {}
    idx = 0
    cases = []
    for n in (4, 10):
        for head in (None, "learnable_sphere", "learnable_box", "fixed_sphere", "fixed_box"):
            cases.append((n, head, [3 * n, 5 * n, 3 * n], "full"))
    cases.append((4, None, [40, 200, 200, 200, 200, 40], "full"))       # main_mlp.py:297-307 dims, n=4
    cases.append((10, None, [100, 500, 500, 500, 500, 100], "sub"))     # n=10: big grads subsampled
    for (n, head, hidden, mode) in cases:
        f = ref_encoders.get_mlp(n_in=n, n_out=n, layers=list(hidden), output_normalization=head)
        fill_formula(f)
        M = 48
        x = f32(formula_weights((M, n), 99) * np.sqrt(n) * 1.5)
        gy = f32(formula_weights((M, n), 77) * np.sqrt(n))
        xt = torch.tensor(x, requires_grad=True)
        y = f(xt)
        y.backward(torch.tensor(gy))
        ins = dict(x=x, gy=gy)
        outs = dict(y=t2n(y), dx=t2n(xt.grad))
        for name, prm in f.named_parameters():
            g = t2n(prm.grad)
            if mode == "sub" and g.size > 20000:
                outs[f"gradsub/{name}"] = subsample(g)
                outs[f"gradsum/{name}"] = np.asarray([g.astype(np.float64).sum(),
                                                      (g.astype(np.float64) ** 2).sum()])
            else:
                outs[f"grad/{name}"] = g
        put(store, f"c{idx:03d}", ins, outs, n=n, head=str(head), hidden=np.asarray(hidden))
        store[f"c{idx:03d}/meta/state_keys"] = np.asarray(list(f.state_dict().keys()))
        idx += 1
    store["n_cases"] = np.asarray(idx)
    save("g6_mlp.npz", store)


# ----------------------------------------------------------------------------- G7
def g7():
    """Five injected unsupervised train steps (main_mlp.py:258-285), B=64, n=4."""
    store = {}
    for ci, (p, head) in enumerate([(2, None), (1, "learnable_sphere")]):
        n, B = 4, 64
        hidden = [n * 10, n * 50, n * 50, n * 50, n * 50, n * 10]
        f = ref_encoders.get_mlp(n_in=n, n_out=n, layers=list(hidden), output_normalization=head)
        fill_formula(f)
        gW = [f32(formula_weights((n, n), 50 + i) * np.sqrt(n)) for i in range(3)]
        g = torch.nn.Sequential(
            torch.nn.Linear(n, n, bias=False), torch.nn.LeakyReLU(0.2),
            torch.nn.Linear(n, n, bias=False), torch.nn.LeakyReLU(0.2),
            torch.nn.Linear(n, n, bias=False))
        for i, li in enumerate((0, 2, 4)):
            g[li].weight.data = torch.tensor(gW[i])
        for prm in g.parameters():
            prm.requires_grad = False
        h = lambda z: f(g(z))  # noqa: E731
        opt = torch.optim.Adam(f.parameters(), lr=1e-3)
        L = ref_losses.LpSimCLRLoss(p=p, tau=1.0, simclr_compatibility_mode=True)
        key = f"c{ci:03d}"
        for i in range(3):
            store[f"{key}/in/g{i}"] = gW[i]
        loss_vals, pos_vals, neg_vals = [], [], []
        for s in range(5):
            z1 = (formula_weights((B, n), 200 + s) * np.sqrt(n) * 0.5 + 0.5).astype(np.float32)
            z2 = np.clip(z1 + 0.05 * formula_weights((B, n), 300 + s) * np.sqrt(n), 0, 1).astype(np.float32)
            store[f"{key}/in/z1_{s}"] = z1
            store[f"{key}/in/z2_{s}"] = z2
            opt.zero_grad()
            a = h(torch.tensor(z1))
            b = h(torch.tensor(z2))
            c = torch.roll(a, 1, 0)
            tot, _, (pm, nm) = L(None, None, None, a, b, c)
            tot.backward()
            opt.step()
            loss_vals.append(tot.item()); pos_vals.append(pm.item()); neg_vals.append(nm.item())
        store[f"{key}/out/loss"] = np.asarray(loss_vals, np.float64)
        store[f"{key}/out/pos"] = np.asarray(pos_vals, np.float64)
        store[f"{key}/out/neg"] = np.asarray(neg_vals, np.float64)
        for name, prm in f.named_parameters():
            w = t2n(prm)
            store[f"{key}/out/param5/{name}"] = w if (ci == 0 or w.size < 2000) else subsample(w, 7)
        store[f"{key}/meta/p"] = np.asarray(p)
        store[f"{key}/meta/head"] = np.asarray(str(head))
        store[f"{key}/meta/lr"] = np.asarray(1e-3)
        store[f"{key}/meta/hidden"] = np.asarray(hidden)
    store["n_cases"] = np.asarray(2)
    save("g7_trainstep.npz", store)


# ----------------------------------------------------------------------------- G8
def g8():
    store = {}
    buf = io.StringIO()
    np.random.seed(0)
    with contextlib.redirect_stdout(buf):
        g = ref_inu.construct_invertible_mlp(n=10, n_layers=3, act_fct="leaky_relu",
                                             cond_thresh_ratio=0.0, n_iter_cond_thresh=25000)
    lines = buf.getvalue().strip().splitlines()
    thresh = float(lines[0].split(":")[1])
    Ws = [t2n(m.weight) for m in g if isinstance(m, torch.nn.Linear)]
    conds = [float(np.linalg.cond(w.astype(np.float64))) for w in Ws]
    x = f32(formula_weights((32, 10), 5) * np.sqrt(10))
    y = t2n(g(torch.tensor(x)))
    for i, w in enumerate(Ws):
        store[f"W{i}"] = w
    store["thresh"] = np.asarray(thresh)
    store["conds"] = np.asarray(conds)
    store["x"] = x
    store["y"] = y
    # a small fast KAT for the constructor itself (n=4, pool of 500)
    np.random.seed(3)
    with contextlib.redirect_stdout(io.StringIO()) as b2:
        g2_ = ref_inu.construct_invertible_mlp(n=4, n_layers=2, act_fct="leaky_relu",
                                               cond_thresh_ratio=0.25, n_iter_cond_thresh=500)
    store["small_thresh"] = np.asarray(float(b2.getvalue().splitlines()[0].split(":")[1]))
    for i, w in enumerate([t2n(m.weight) for m in g2_ if isinstance(m, torch.nn.Linear)]):
        store[f"small_W{i}"] = w
    save("g8_mixing.npz", store)


# ----------------------------------------------------------------------------- G9
def g9():
    """Sampler statistics (distributional parity only; RNG streams are not reproducible on device)."""
    store = {}
    N = 100000
    qs = np.asarray([0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99])
    torch.manual_seed(0)
    np.random.seed(0)
    box = ref_spaces.NBoxSpace(10, 0.0, 1.0)
    z = box.uniform(N)
    zt = box.normal(z, 0.05, N)
    store["box_uniform/mean"] = t2n(z.mean(0)); store["box_uniform/var"] = t2n(z.var(0))
    store["box_uniform/q"] = np.quantile(t2n(z), qs, axis=0).astype(np.float32)
    d = t2n(zt - z)
    store["box_normal/delta_mean"] = d.mean(0); store["box_normal/delta_var"] = d.var(0)
    store["box_normal/delta_q"] = np.quantile(d, qs, axis=0).astype(np.float32)
    store["box_normal/min"] = np.asarray(t2n(zt).min()); store["box_normal/max"] = np.asarray(t2n(zt).max())
    # fixed mean near the boundary: the truncated-normal shape is what matters
    mean_edge = torch.full((N, 10), 0.02)
    ze = t2n(box.normal(mean_edge, 0.05, N))
    store["box_normal_edge/mean"] = ze.mean(0); store["box_normal_edge/var"] = ze.var(0)
    store["box_normal_edge/q"] = np.quantile(ze, qs, axis=0).astype(np.float32)
    zl = t2n(box.laplace(mean_edge, 0.05, N))
    store["box_laplace_edge/mean"] = zl.mean(0); store["box_laplace_edge/var"] = zl.var(0)
    store["box_laplace_edge/q"] = np.quantile(zl, qs, axis=0).astype(np.float32)
    zg = t2n(box.generalized_normal(torch.full((N, 10), 0.5), 0.05, p=3, size=N))
    store["box_gennorm3/mean"] = zg.mean(0); store["box_gennorm3/var"] = zg.var(0)
    store["box_gennorm3/q"] = np.quantile(zg, qs, axis=0).astype(np.float32)

    sph = ref_spaces.NSphereSpace(10)
    s = sph.uniform(N)
    st = sph.normal(s, 0.05, N)
    store["sphere_uniform/mean"] = t2n(s.mean(0)); store["sphere_uniform/var"] = t2n(s.var(0))
    store["sphere_uniform/norm_err"] = np.asarray(float((s.norm(dim=-1) - 1).abs().max()))
    cosang = t2n((s * st).sum(-1))
    store["sphere_normal/cos_mean"] = np.asarray(cosang.mean()); store["sphere_normal/cos_var"] = np.asarray(cosang.var())
    store["sphere_normal/cos_q"] = np.quantile(cosang, qs).astype(np.float32)
    sl = sph.laplace(s, 0.05, N)
    cl = t2n((s * sl).sum(-1))
    store["sphere_laplace/cos_mean"] = np.asarray(cl.mean()); store["sphere_laplace/cos_q"] = np.quantile(cl, qs).astype(np.float32)
    for kappa in (1.0, 10.0, 100.0):
        mu = torch.zeros(10); mu[0] = 1.0
        v = sph.von_mises_fisher(mu, kappa, N)
        c = t2n(v[:, 0])
        store[f"vmf_k{int(kappa)}/cos_mean"] = np.asarray(c.mean())
        store[f"vmf_k{int(kappa)}/cos_var"] = np.asarray(c.var())
        store[f"vmf_k{int(kappa)}/cos_q"] = np.quantile(c, qs).astype(np.float32)
        store[f"vmf_k{int(kappa)}/norm_err"] = np.asarray(float((v.norm(dim=-1) - 1).abs().max()))
        store[f"vmf_k{int(kappa)}/orth_var"] = t2n(v[:, 1:].var(0))
    real = ref_spaces.NRealSpace(10)
    rn = t2n(real.normal(torch.zeros(10), 2.0, N)); store["real_normal/var"] = rn.var(0)
    rl = t2n(real.laplace(torch.zeros(10), 0.7, N)); store["real_laplace/var"] = rl.var(0)
    store["real_laplace/q"] = np.quantile(rl, qs, axis=0).astype(np.float32)
    rg = t2n(real.generalized_normal(torch.zeros(1, 10), 0.7, p=3, size=N))
    store["real_gennorm3/var"] = rg.var(0); store["real_gennorm3/q"] = np.quantile(rg, qs, axis=0).astype(np.float32)
    store["quantiles"] = qs
    save("g9_samplers.npz", store)


# ----------------------------------------------------------------------------- G10
def g10():
    store = {}
    # KITTI pattern: mu[::2], mu[1::2] (kitti_masks/solver.py:64-65)
    g = torch.Generator().manual_seed(77)
    mu = torch.randn(64, 5, generator=g)
    mu_l = mu.clone().requires_grad_(True)
    a, b = mu_l[::2], mu_l[1::2]
    L = ref_losses.LpSimCLRLoss(p=1, tau=1.0, simclr_compatibility_mode=True)
    tot, per, (pm, nm) = L(None, None, None, a, b, torch.roll(a, 1, 0))
    tot.backward()
    put(store, "kitti", dict(mu=t2n(mu)),
        dict(loss_mean=t2n(tot), loss_i=t2n(per), pos_mean=t2n(pm), neg_mean=t2n(nm), dmu=t2n(mu_l.grad)),
        p=1, tau=1.0)
    # 3DIdent pattern: column slices z[:, :k] (main_3dident.py:429-438)
    zz = torch.randn(32, 10, generator=g)
    z2 = zz + 0.1 * torch.randn(32, 10, generator=g)
    za = zz.clone().requires_grad_(True); zb = z2.clone().requires_grad_(True)
    L = ref_losses.LpSimCLRLoss(p=2, tau=1.0, simclr_compatibility_mode=True)
    tot, per, (pm, nm) = L(None, None, None, za[:, :3], zb[:, :3], torch.roll(za, 1, 0)[:, :3])
    tot.backward()
    put(store, "ident", dict(z=t2n(zz), z2=t2n(z2)),
        dict(loss_mean=t2n(tot), loss_i=t2n(per), dz=t2n(za.grad), dz2=t2n(zb.grad)), p=2, tau=1.0, k=3)
    save("g10_strided.npz", store)


# ----------------------------------------------------------------------------- G11
def g11():
    """Evaluation metrics (disentanglement_utils.py): the two calls main_mlp.py makes."""
    import disentanglement_utils as ref_du
    store = {}
    rng = np.random.default_rng(0)
    for i, (N, n) in enumerate([(2048, 10), (1000, 4), (512, 40)]):
        z = rng.uniform(size=(N, n)).astype(np.float32)
        A = rng.normal(size=(n, n))
        hz = (np.tanh(z @ A) + 0.05 * rng.normal(size=(N, n))).astype(np.float32)[:, rng.permutation(n)]
        (r2, _), _ = ref_du.linear_disentanglement(torch.tensor(z), torch.tensor(hz), mode="r2")
        (mcc, corr), _ = ref_du.permutation_disentanglement(torch.tensor(z), torch.tensor(hz), mode="pearson",
                                                            solver="munkres", rescaling=True)
        store[f"c{i}/z"] = z; store[f"c{i}/hz"] = hz
        store[f"c{i}/r2"] = np.asarray(r2); store[f"c{i}/mcc"] = np.asarray(mcc); store[f"c{i}/corr_diag"] = np.diag(corr)
    store["n_cases"] = np.asarray(3)
    save("g11_metrics.npz", store)


def g12():
    """UniformityLoss / AlignmentLoss (losses.py:205-241).  (AlignmentUniformityLoss raises inside the reference's
    SplitCombinedCLLoss -- torch.tensor() of a list of (B,) tensors, losses.py:144 -- so there is nothing to pin.)"""
    store = {}
    idx = 0
    for p in (1.0, 2.0, 3.0):
        for (B, B3, n, scale) in ((8, 8, 3, 1.0), (64, 48, 10, 1.0), (96, 96, 10, 0.3)):
            z1, z2, z3 = rand_inputs(12000 + idx, B, B3, n, scale)
            a = torch.tensor(z1, requires_grad=True); c = torch.tensor(z3, requires_grad=True)
            u, ui, _ = ref_losses.UniformityLoss(p)(a, c)
            u.backward()
            put(store, f"u{idx:03d}", dict(z1=z1, z3=z3), dict(loss_mean=t2n(u), loss_i=t2n(ui), dz1=t2n(a.grad), dz3=t2n(c.grad)), p=p)
            a = torch.tensor(z1, requires_grad=True); b = torch.tensor(z2, requires_grad=True)
            al, ali, _ = ref_losses.AlignmentLoss(p)(a, b)
            al.backward()
            put(store, f"a{idx:03d}", dict(z1=z1, z2=z2), dict(loss_mean=t2n(al), loss_i=t2n(ali), dz1=t2n(a.grad), dz2=t2n(b.grad)), p=p)
            idx += 1
    store["n_cases"] = np.asarray(idx)
    save("g12_align_uniform.npz", store)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g1r", "g2", "g3", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12"]
    table = dict(g1=g1, g1r=g1_roll, g2=g2, g3=g3, g5=g5, g6=g6, g7=g7, g8=g8, g9=g9, g10=g10, g11=g11, g12=g12)
    for w in which:
        table[w]()
