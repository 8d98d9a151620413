"""HIP loss kernels (through the C ABI / ctypes) against the reference goldens and the oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-5   # north_star: loss value and embedding gradients within 1e-5 relative fp32


def dev(a):
    return torch.tensor(np.asarray(a, np.float32), device="cuda")


def grad_scale(z1, z2, p, tau, alpha):
    d = np.abs(np.asarray(z1, np.float64) - np.asarray(z2, np.float64))
    return 2 * alpha / (d.shape[0] * tau) * float((p * np.maximum(d, 1e-12) ** (p - 1)).max())


def run_hip(loss_obj, z1, z2, z3, roll=False):
    a = dev(z1).requires_grad_(True)
    b = dev(z2).requires_grad_(True)
    if roll:
        c = torch.roll(a, 1, 0)
    else:
        c = dev(z3).requires_grad_(True)
    mean, per, (pm, nm) = loss_obj(None, None, None, a, b, c)
    mean.backward()
    out = dict(loss_mean=mean.item(), loss_i=per.detach().cpu().numpy(), pos_mean=pm.item(), neg_mean=nm.item(),
               dz1=a.grad.cpu().numpy(), dz2=b.grad.cpu().numpy())
    if not roll:
        out["dz3"] = c.grad.cpu().numpy()
    return out


def compare(out, ref, lse_scale, gscale, grads, tol=TOL):
    comp = max(float(np.abs(ref["loss_i"]).max()), lse_scale, 1e-30)
    assert abs(out["loss_mean"] - float(ref["loss_mean"])) < tol * comp
    assert np.abs(out["loss_i"] - ref["loss_i"]).max() < tol * comp
    if "pos_mean" in ref:
        assert abs(out["pos_mean"] - float(ref["pos_mean"])) < tol * max(1.0, abs(float(ref["pos_mean"])))
        assert abs(out["neg_mean"] - float(ref["neg_mean"])) < tol * max(1.0, abs(float(ref["neg_mean"])), lse_scale)
    sat = max(1.0, lse_scale)
    for g in grads:
        scale = max(float(np.abs(ref[g]).max()), gscale, 1e-30)
        assert np.abs(out[g] - ref[g]).max() / scale < tol * sat, g


@pytest.mark.parametrize("name", ["g1_lp_loss.npz", "g2_rect.npz", "g3_misc.npz"])
def test_lp_goldens(golden, name):
    from cl_ica_amd.losses import LpSimCLRLoss
    G = golden(name)
    for key, c in G.cases():
        m = c["meta"]
        p = float(m["p"]); p = int(p) if p == int(p) else p
        L = LpSimCLRLoss(p=p, tau=float(m["tau"]), alpha=float(m["alpha"]),
                         simclr_compatibility_mode=bool(m["compat"]), pow=bool(m["pow"]))
        out = run_hip(L, c["in"]["z1"], c["in"]["z2"], c["in"]["z3"])
        orc = O.lp_simclr_loss(c["in"]["z1"], c["in"]["z2"], c["in"]["z3"], p=float(m["p"]), tau=float(m["tau"]),
                               alpha=float(m["alpha"]), compat=bool(m["compat"]), pow=bool(m["pow"]), grad=False)
        lse_scale = float(np.abs(orc["lse"]).max()) + np.log(c["in"]["z3"].shape[0] + 1.0)
        gs = grad_scale(c["in"]["z1"], c["in"]["z2"], float(m["p"]), float(m["tau"]), float(m["alpha"]))
        try:
            compare(out, c["out"], lse_scale, gs, ("dz1", "dz2", "dz3"))
        except AssertionError as e:
            raise AssertionError(f"{name}:{key} meta={ {k: v.tolist() for k, v in m.items()} }: {e}")


def test_lp_roll_goldens(golden):
    """z3 = roll(z1) inside the autograd graph (main_mlp.py:272): exact-zero distances every row."""
    from cl_ica_amd.losses import LpSimCLRLoss
    for key, c in golden("g1r_lp_roll.npz").cases():
        m = c["meta"]
        L = LpSimCLRLoss(p=int(m["p"]), tau=float(m["tau"]), simclr_compatibility_mode=True)
        out = run_hip(L, c["in"]["z1"], c["in"]["z2"], None, roll=True)
        z1 = c["in"]["z1"]
        orc = O.lp_simclr_loss(z1, c["in"]["z2"], np.roll(z1, 1, 0), p=float(m["p"]), compat=True, grad=False)
        lse_scale = float(np.abs(orc["lse"]).max()) + np.log(z1.shape[0] + 1.0)
        gs = grad_scale(z1, c["in"]["z2"], float(m["p"]), 1.0, 0.5)
        assert not np.isnan(out["dz1"]).any()
        compare(out, c["out"], lse_scale, gs, ("dz1", "dz2"))


def test_simclr_goldens(golden):
    from cl_ica_amd.losses import SimCLRLoss
    for key, c in golden("g5_simclr.npz").cases():
        m = c["meta"]
        L = SimCLRLoss(normalize=bool(m["normalize"]), tau=float(m["tau"]), alpha=float(m["alpha"]))
        out = run_hip(L, c["in"]["z1"], c["in"]["z2"], c["in"]["z3"])
        ref = c["out"]
        lse_scale = float(np.abs(ref["neg_mean"])) + np.log(c["in"]["z3"].shape[0] + 1.0)
        gscale = float(max(np.abs(c["in"]["z1"]).max(), 1.0)) / (c["in"]["z1"].shape[0] * float(m["tau"]))
        compare(out, ref, lse_scale, gscale, ("dz1", "dz2", "dz3"))


def test_strided_views(golden):
    """mu[::2] / mu[1::2] (kitti_masks/solver.py:64-65) and z[:, :k] (main_3dident.py:429-438)."""
    from cl_ica_amd.losses import LpSimCLRLoss
    G = golden("g10_strided.npz")
    c = G.case("kitti")
    mu = dev(c["in"]["mu"]).requires_grad_(True)
    a, b = mu[::2], mu[1::2]
    tot, per, (pm, nm) = LpSimCLRLoss(p=1, tau=1.0, simclr_compatibility_mode=True)(None, None, None, a, b, torch.roll(a, 1, 0))
    tot.backward()
    assert abs(tot.item() - float(c["out"]["loss_mean"])) < TOL * 8
    assert rel_err(per.detach().cpu().numpy(), c["out"]["loss_i"]) < TOL
    assert np.abs(mu.grad.cpu().numpy() - c["out"]["dmu"]).max() < TOL * max(np.abs(c["out"]["dmu"]).max(), 2.0 / 32)
    c = G.case("ident")
    za = dev(c["in"]["z"]).requires_grad_(True); zb = dev(c["in"]["z2"]).requires_grad_(True)
    tot, per, _ = LpSimCLRLoss(p=2, tau=1.0, simclr_compatibility_mode=True)(
        None, None, None, za[:, :3], zb[:, :3], torch.roll(za, 1, 0)[:, :3])
    tot.backward()
    assert rel_err(per.detach().cpu().numpy(), c["out"]["loss_i"]) < TOL
    assert np.abs(za.grad.cpu().numpy() - c["out"]["dz"]).max() < TOL * max(np.abs(c["out"]["dz"]).max(), 1e-3)
    assert np.abs(zb.grad.cpu().numpy() - c["out"]["dz2"]).max() < TOL * max(np.abs(c["out"]["dz2"]).max(), 1e-3)


def test_analytic_kats():
    from cl_ica_amd.losses import LpSimCLRLoss
    for B in (8, 512, 6144):
        z = torch.zeros(B, 10, device="cuda")
        v = LpSimCLRLoss(p=2, simclr_compatibility_mode=True)(None, None, None, z, z, z)[0].item()
        assert abs(v - np.log(B + 1)) < 1e-5 * np.log(B + 1)
        v = LpSimCLRLoss(p=2)(None, None, None, z, z, z)[0].item()
        assert abs(v) < 1e-5


@pytest.mark.parametrize("B,B3,n,p", [(6144, 6144, 10, 2), (6144, 6144, 10, 1), (1000, 3001, 7, 3), (6144, 12288, 40, 1),
                                       (300, 5000, 33, 2), (257, 63, 1, 2), (2048, 2048, 64, 1.5), (1, 1, 3, 2), (1, 777, 5, 1),
                                       (5, 2, 2, 3)])
def test_full_size_vs_oracle(B, B3, n, p):
    """BASELINE sizes and ragged shapes vs the fp64 oracle; all four upstream gradients exercised."""
    from cl_ica_amd.losses import LpSimCLRLoss
    rng = np.random.default_rng(B + n)
    z1 = rng.normal(size=(B, n)).astype(np.float32) * 0.7
    z2 = (z1 + 0.05 * rng.normal(size=(B, n))).astype(np.float32)
    z3 = rng.normal(size=(B3, n)).astype(np.float32) * 0.7
    gi = (rng.normal(size=B) / B).astype(np.float32)
    for compat in (True, False):
        a = dev(z1).requires_grad_(True); b = dev(z2).requires_grad_(True); c = dev(z3).requires_grad_(True)
        mean, per, (pm, nm) = LpSimCLRLoss(p=p, tau=0.8, alpha=0.4, simclr_compatibility_mode=compat)(None, None, None, a, b, c)
        (1.3 * mean + (per * dev(gi)).sum() + 0.4 * pm - 0.2 * nm).backward()
        orc = O.lp_simclr_loss(z1, z2, z3, p=p, tau=0.8, alpha=0.4, compat=compat, g_mean=1.3, g_item=gi, g_pos=0.4, g_neg=-0.2)
        comp = float(np.abs(orc["lse"]).max()) + np.log(B3 + 1.0)
        assert abs(mean.item() - orc["loss_mean"]) < TOL * comp
        assert np.abs(per.detach().cpu().numpy() - orc["loss_i"]).max() < TOL * comp
        assert abs(pm.item() - orc["pos_mean"]) < TOL * max(1.0, abs(orc["pos_mean"]))
        assert abs(nm.item() - orc["neg_mean"]) < TOL * comp
        gs = grad_scale(z1, z2, p, 0.8, 0.4) * 2.5
        for got, name in ((a.grad, "dz1"), (b.grad, "dz2"), (c.grad, "dz3")):
            ref = orc[name]
            scale = max(np.abs(ref).max(), gs if name != "dz3" else 0.0, 1e-30)
            assert np.abs(got.cpu().numpy() - ref).max() / scale < TOL * max(1.0, comp), (name, compat)


def test_permutation_invariance_and_roll_identity():
    """Size-independent properties at BASELINE size: the LSE is invariant to the order of z3 rows, so
    z3 = roll(z1) and z3 = z1 give the same loss (what the fused train step relies on)."""
    from cl_ica_amd.losses import LpSimCLRLoss
    torch.manual_seed(0)
    a = torch.randn(6144, 10, device="cuda"); b = a + 0.05 * torch.randn_like(a)
    L = LpSimCLRLoss(p=2, tau=1.0, simclr_compatibility_mode=True)
    v1 = L(None, None, None, a, b, torch.roll(a, 1, 0))
    v2 = L(None, None, None, a, b, a)
    perm = torch.randperm(6144, device="cuda")
    v3 = L(None, None, None, a, b, a[perm])
    assert abs(v1[0].item() - v2[0].item()) < 1e-6 * abs(v1[0].item())
    assert abs(v1[0].item() - v3[0].item()) < 1e-6 * abs(v1[0].item())
    assert torch.allclose(v1[1], v3[1], rtol=1e-5, atol=1e-6)


def test_error_paths():
    from cl_ica_amd.losses import LpSimCLRLoss
    from cl_ica_amd._lib import ClicaError
    z = torch.zeros(4, 3)
    with pytest.raises(ClicaError):
        LpSimCLRLoss(p=2)(None, None, None, z, z, z)          # CPU tensors: no fallback
    zc = torch.zeros(4, 70, device="cuda")
    with pytest.raises(ClicaError):
        LpSimCLRLoss(p=2)(None, None, None, zc, zc, zc)       # n > 64
    with pytest.raises(ClicaError):
        LpSimCLRLoss(p=0.5)(None, None, None, zc[:, :3], zc[:, :3], torch.zeros(5, 3, device="cuda"))


def test_rowgrad_forward_equals_row_pass():
    """The flash-style row gradient accumulated in the forward sweep equals the backward's recomputing
    row pass (C ABI called directly both ways, all exponent kinds + dot)."""
    import ctypes as C
    from cl_ica_amd import _lib
    lib = _lib.load()
    torch.manual_seed(3)
    B, B3, n = 1500, 2100, 10
    z1 = torch.randn(B, n, device="cuda") * 0.6; z2 = z1 + 0.05 * torch.randn_like(z1); z3 = torch.randn(B3, n, device="cuda") * 0.6
    for p, pw in ((1, 1), (2, 1), (3, 1), (1.5, 1), (2, 0)):
        d = _lib.LpLossDesc(B=B, B3=B3, n=n, p=float(p), tau=0.9, alpha=0.4, compat=1, pow=pw)
        fb, bb = C.c_size_t(), C.c_size_t()
        _lib.check(lib.clica_lp_loss_workspace_bytes(C.byref(d), C.byref(fb), C.byref(bb)), "ws")
        ws = torch.zeros(max(fb.value, bb.value), dtype=torch.uint8, device="cuda")
        outs = []
        for use_rg in (False, True):
            o = torch.empty(3 * B + 3, device="cuda"); rg = torch.empty(B, n, device="cuda")
            dz = [torch.empty(B, n, device="cuda"), torch.empty(B, n, device="cuda"), torch.empty(B3, n, device="cuda")]
            rgp = rg.data_ptr() if use_rg else None
            _lib.check(lib.clica_lp_loss_fwd(C.byref(d), z1.data_ptr(), n, z2.data_ptr(), n, z3.data_ptr(), n, o[:B].data_ptr(),
                                             o[B:2 * B].data_ptr(), o[2 * B:3 * B].data_ptr(), o[3 * B:].data_ptr(), rgp, n,
                                             ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "fwd")
            _lib.check(lib.clica_lp_loss_bwd(C.byref(d), z1.data_ptr(), n, z2.data_ptr(), n, z3.data_ptr(), n, o[2 * B:3 * B].data_ptr(),
                                             rgp, n, None, None, None, None, dz[0].data_ptr(), n, dz[1].data_ptr(), n, dz[2].data_ptr(), n, 0,
                                             ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "bwd")
            torch.cuda.synchronize()
            outs.append((o.clone(), [t.clone() for t in dz]))
        assert torch.allclose(outs[0][0], outs[1][0], rtol=2e-6, atol=1e-6), p
        for a, b in zip(outs[0][1], outs[1][1]):
            assert (a - b).abs().max().item() < 2e-6 * max(a.abs().max().item(), 1e-6) + 1e-9, p


def test_uniformity_alignment_vs_golden(golden):
    """UniformityLoss / AlignmentLoss (reference losses.py:205-241) on the all-pairs kernel, against G12."""
    from cl_ica_amd.losses import AlignmentLoss, UniformityLoss
    G = golden("g12_align_uniform.npz")
    for i in range(G.n_cases):
        u = G.case(f"u{i:03d}"); p = float(u["meta"]["p"])
        z1 = dev(u["in"]["z1"]).requires_grad_(True); z3 = dev(u["in"]["z3"]).requires_grad_(True)
        tot, per, extra = UniformityLoss(p)(z1, z3)
        tot.backward()
        comp = max(float(np.abs(u["out"]["loss_i"]).max()) + np.log(z1.shape[0]), 1.0)
        assert abs(tot.item() - float(u["out"]["loss_mean"])) < TOL * comp and extra[0] is tot
        assert np.abs(per.detach().cpu().numpy() - u["out"]["loss_i"]).max() < TOL * comp
        for name, t in (("dz1", z1), ("dz3", z3)):
            d = np.abs(np.asarray(u["in"]["z1"], np.float64)[None] - np.asarray(u["in"]["z3"], np.float64)[:, None])
            scale = max(np.abs(u["out"][name]).max(), float((p * np.maximum(d, 1e-12) ** (p - 1)).max()) / z3.shape[0])
            assert np.abs(t.grad.cpu().numpy() - u["out"][name]).max() / scale < TOL * comp, (i, name)
        a = G.case(f"a{i:03d}")
        z1 = dev(a["in"]["z1"]).requires_grad_(True); z2 = dev(a["in"]["z2"]).requires_grad_(True)
        tot, per, _ = AlignmentLoss(p)(z1, z2)
        tot.backward()
        assert abs(tot.item() - float(a["out"]["loss_mean"])) < TOL * max(1.0, abs(float(a["out"]["loss_mean"])))
        assert rel_err(per.detach().cpu().numpy(), a["out"]["loss_i"]) < TOL
        assert rel_err(z1.grad.cpu().numpy(), a["out"]["dz1"]) < TOL and rel_err(z2.grad.cpu().numpy(), a["out"]["dz2"]) < TOL
    with pytest.raises(NotImplementedError):
        UniformityLoss(0.5)(z1, z2)
