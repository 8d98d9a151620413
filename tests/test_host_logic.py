"""Host-side logic that needs no GPU: module structure / state-dict compatibility, mixing-network
construction KAT, fail-loud behaviour on CPU tensors, arena flattening, gradient bucketing."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from conftest import ROOT as ROOT_DIR, rel_err


def test_get_mlp_structure_and_state_dict(golden):
    from cl_ica_amd import encoders, layers
    G = golden("g6_mlp.npz")
    for key, c in G.cases():
        n = int(c["meta"]["n"]); head = str(c["meta"]["head"]); head = None if head == "None" else head
        hidden = [int(h) for h in c["meta"]["hidden"]]
        arg = list(hidden)
        f = encoders.get_mlp(n_in=n, n_out=n, layers=arg, output_normalization=head)
        assert arg == hidden + [n]                      # the reference appends n_out in place (encoders.py:56)
        assert isinstance(f, torch.nn.Sequential)
        assert list(f.state_dict().keys()) == [str(k) for k in c["meta"]["state_keys"]]
        dims = [n] + hidden + [n]
        lin = [m for m in f if isinstance(m, torch.nn.Linear)]
        assert [(m.in_features, m.out_features) for m in lin] == list(zip(dims[:-1], dims[1:]))
        assert all(m.negative_slope == 0.01 for m in f if isinstance(m, torch.nn.LeakyReLU))
        if head == "learnable_sphere":
            assert isinstance(f[-1], layers.RescaleLayer) and isinstance(f[-1].r, torch.nn.Parameter) and f[-1].r.shape == (1,)
        if head == "fixed_sphere":
            assert not isinstance(f[-1].r, torch.nn.Parameter) and "r" not in dict(f[-1].named_buffers())
        if head == "learnable_box":
            assert f[-1].max_abs_bound.shape == (n,)
        # nn.Linear default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight (kaiming a=sqrt(5)) and bias
        for m in lin:
            bound = 1.0 / np.sqrt(m.in_features)
            assert float(m.weight.abs().max()) <= bound + 1e-6 and float(m.bias.abs().max()) <= bound + 1e-6


def test_get_mlp_rejects_unbuilt_options():
    from cl_ica_amd import encoders
    f = encoders.get_mlp(4, 4, [8, 6], layer_normalization="bn")       # reference module order: Linear, norm, activation
    assert [type(m).__name__ for m in f] == ["Linear", "BatchNorm1d", "LeakyReLU", "Linear", "BatchNorm1d", "LeakyReLU", "Linear"]
    assert isinstance(encoders.get_mlp(4, 4, [8], layer_normalization="gn")[1], torch.nn.GroupNorm)
    with pytest.raises(ValueError):
        encoders.get_mlp(4, 4, [8], layer_normalization="ln")
    with pytest.raises(ValueError):
        encoders.get_mlp(4, 4, [8], output_normalization="nope")
    with pytest.raises(ValueError):
        encoders.get_mlp(4, 4, [])


def test_mixing_constructor_kat(golden):
    """np.random.seed(0), n=10, L=3 reproduces the reference's g bit-for-bit."""
    from cl_ica_amd import invertible_network_utils as inu
    z = golden("g8_mixing.npz").z
    np.random.seed(3)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g = inu.construct_invertible_mlp(n=4, n_layers=2, act_fct="leaky_relu", cond_thresh_ratio=0.25, n_iter_cond_thresh=500)
    assert abs(float(buf.getvalue().splitlines()[0].split(":")[1]) - float(z["small_thresh"])) < 1e-6
    Ws = [m.weight.detach().numpy() for m in g if isinstance(m, torch.nn.Linear)]
    assert np.array_equal(Ws[0], z["small_W0"]) and np.array_equal(Ws[1], z["small_W1"])
    assert list(g.state_dict().keys()) == ["0.weight", "2.weight"]
    assert all(not p.requires_grad for p in g.parameters())
    assert isinstance(g[1], torch.nn.LeakyReLU) and g[1].negative_slope == 0.2
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()) as b2:
        g = inu.construct_invertible_mlp(n=10, n_layers=3, act_fct="leaky_relu", cond_thresh_ratio=0.0, n_iter_cond_thresh=25000)
    assert abs(float(b2.getvalue().splitlines()[0].split(":")[1]) - 4.085284) < 1e-6
    for i, m in enumerate([m for m in g if isinstance(m, torch.nn.Linear)]):
        assert np.array_equal(m.weight.detach().numpy(), z[f"W{i}"])
    with pytest.raises(NotImplementedError):       # the reference raises for max_out as well (invertible_network_utils.py:58-59)
        inu.construct_invertible_mlp(n=4, n_layers=2, act_fct="max_out", n_iter_cond_thresh=10)
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        gs = inu.construct_invertible_mlp(n=4, n_layers=2, act_fct="softplus", n_iter_cond_thresh=10)
    assert isinstance(gs[1], torch.nn.Softplus) and gs.act_kind == 3 and list(gs.state_dict().keys()) == ["0.weight", "2.weight"]


def test_no_cpu_fallback_anywhere():
    """CPU tensors must raise, never silently compute (the oracle is test infrastructure only)."""
    from cl_ica_amd import encoders, layers, losses, spaces
    from cl_ica_amd._lib import ClicaError
    z = torch.randn(8, 3)
    with pytest.raises(ClicaError):
        losses.LpSimCLRLoss(p=2)(None, None, None, z, z, z)
    with pytest.raises(ClicaError):
        losses.SimCLRLoss()(None, None, None, z, z, z)
    with pytest.raises(ClicaError):
        encoders.get_mlp(3, 3, [6])(z)
    with pytest.raises(ClicaError):
        layers.RescaleLayer()(z)
    with pytest.raises(RuntimeError):
        spaces.NSphereSpace(3).uniform(5, device="cpu")
    import cl_ica_amd
    import glob
    srcs = glob.glob(f"{cl_ica_amd.__path__[0]}/**/*.py", recursive=True)
    assert len(srcs) >= 15
    for path in srcs:      # every module of the package, sub-packages included
        src = open(path).read()
        assert "import oracle" not in src and "from oracle" not in src, path


def test_loss_ctor_surface():
    from cl_ica_amd import losses
    L = losses.LpSimCLRLoss(p=1)
    assert (L.p, L.tau, L.alpha, L.simclr_compatibility_mode, L.pow) == (1, 1.0, 0.5, False, True)
    S = losses.SimCLRLoss()
    assert (S.normalize, S.tau, S.alpha) == (False, 1.0, 0.5)
    assert isinstance(L, losses.CLLoss) and isinstance(S, losses.CLLoss)
    with pytest.raises(ValueError):
        L(None, None, None, torch.zeros(4, 3), torch.zeros(5, 3), torch.zeros(4, 3))


def test_latent_space_plumbing():
    from cl_ica_amd import latent_spaces, spaces
    calls = []

    class Fake(spaces.Space):
        dim = 3

        def uniform(self, size, device="cpu"):
            calls.append(("u", size)); return torch.zeros(size, 3)

        def normal(self, mean, std, size, device="cpu"):
            calls.append(("n", std)); return mean + 1

        laplace = generalized_normal = normal

    ls = latent_spaces.LatentSpace(Fake(), lambda sp, size, device="cpu": sp.uniform(size, device),
                                   lambda sp, z, size, device="cpu": sp.normal(z, 0.05, size, device))
    z = ls.sample_marginal(size=5)
    zt = ls.sample_conditional(z, size=5)
    assert calls == [("u", 5), ("n", 0.05)] and ls.dim == 3 and torch.equal(zt, z + 1)
    prod = latent_spaces.ProductLatentSpace([ls, ls])
    assert prod.dim == 6 and prod.sample_marginal(size=2).shape == (2, 6)
    assert prod.sample_conditional(torch.zeros(2, 6), size=2).shape == (2, 6)
    empty = latent_spaces.LatentSpace(Fake(), None, None)
    with pytest.raises(RuntimeError):
        empty.sample_marginal


def test_trainer_arena_flattening_cpu():
    """Arena construction is host logic (no kernel launch): parameters become views of one flat buffer,
    state-dict keys/values are preserved, layer slices tile the arena in backward order."""
    from cl_ica_amd import encoders
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    torch.manual_seed(0)
    f = encoders.get_mlp(4, 4, [40, 200, 40], output_normalization="learnable_sphere")
    before = {k: v.clone() for k, v in f.state_dict().items()}
    tr = ContrastiveTrainer(f, torch.randn(3, 4, 4), SamplerSpec(n=4), batch_size=64, p=1, device="cpu")
    after = f.state_dict()
    assert list(after.keys()) == list(before.keys())
    for k in before:
        assert torch.equal(before[k], after[k])
    base = tr.param_arena.data_ptr()
    for p in f.parameters():
        assert base <= p.data_ptr() < base + tr.param_arena.numel() * 4
        assert (p.data_ptr() - base) % 16 == 0
        assert p.grad is not None and p.grad.shape == p.shape
    tr.param_arena.zero_()
    assert all(float(p.abs().max()) == 0 for p in f.parameters())       # really views
    sl = tr._layer_slices
    assert len(sl) == 4 and sl[0][0] > sl[-1][0]                        # last layer first
    assert all(a < b for a, b in sl)


def test_grad_bucket_partition():
    from cl_ica_amd.distributed import GradBuckets
    arena = torch.zeros(1000)
    slices = [(900, 1000), (600, 900), (100, 600), (0, 100)]     # backward completion order
    gb = GradBuckets(arena, slices, world=2, group=None, bucket_bytes=1200)
    assert gb.buckets == [(600, 1000), (100, 600), (0, 100)]
    assert gb.trigger == {1: 0, 2: 1, 3: 2}
    covered = sorted(gb.buckets)
    assert covered[0][0] == 0 and covered[-1][1] == 1000
    assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))


def _host_moment_stand_ins(monkeypatch):
    """The two device entry points disentanglement_utils calls, stood in for by host arithmetic so that ITS host logic (every score as a
    function of the moment matrix) is tested without a GPU; the kernels themselves are tested with -m gpu."""
    from cl_ica_amd import ops

    def moments(a, b=None):
        cols = [a.double()] + ([b.double()] if b is not None else []) + [torch.ones(a.shape[0], 1, dtype=torch.float64)]
        X = torch.cat(cols, 1)
        return X.T @ X

    def linear_fwd(x, w, bias, leaky, slope=0.01, out=None):
        y = x.double() @ w.double().T
        return (y + bias.double() if bias is not None else y).float()
    monkeypatch.setattr(ops, "moments", moments)
    monkeypatch.setattr(ops, "linear_fwd", linear_fwd)


def test_metrics_goldens(golden, monkeypatch):
    """R^2 / MCC from the moment matrix vs the reference's sklearn + Munkres values (G11)."""
    from cl_ica_amd import disentanglement_utils as du
    _host_moment_stand_ins(monkeypatch)
    z9 = golden("g11_metrics.npz").z
    for i in range(int(z9["n_cases"])):
        z, hz = torch.tensor(z9[f"c{i}/z"]), torch.tensor(z9[f"c{i}/hz"])
        (r2, none), (z2, pred) = du.linear_disentanglement(z, hz, mode="r2")
        assert none is None and pred.shape == z.shape
        assert abs(r2 - float(z9[f"c{i}/r2"])) < 1e-6
        (mcc, corr), thz = du.permutation_disentanglement(z, hz, mode="pearson", solver="munkres", rescaling=True)
        assert abs(mcc - float(z9[f"c{i}/mcc"])) < 1e-6
        assert np.abs(np.abs(np.diag(corr)) - np.abs(z9[f"c{i}/corr_diag"])).max() < 1e-6
        assert thz.shape == z.shape


def check_metric_modes(du, z25, to_t, tol=2e-6, check=None):
    """Every mode / solver of disentanglement_utils.py:63-221 against G25 (tests/golden/gen_goldens_r4.py); shared with the GPU test."""
    def close(a, b, what):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        # rank statistics are discontinuous in the values ranked: where the ranked quantity is itself COMPUTED (the regression's
        # prediction, in fp32 here and in the reference) two near-equal predictions may swap ranks
        # (the reference's own value carries that noise: sklearn fits in fp32); a swap moves the coefficient by O(1 / N)
        # rescaling=True ranks fp32-ROUNDED products hz * beta in the reference (numpy float32 matmul with the diagonal matrix): the
        # same effect at a smaller scale
        t = 0.1 / rows_eval[0] if ("spearman" in what and "/lin/" in what) else (0.02 / z.shape[0] if "spearman" in what else tol)
        if check is not None:
            check(what, a, b, t)
        else:
            assert a.shape == b.shape and np.abs(a - b).max() <= t * max(1.0, np.abs(b).max()), (what, float(np.abs(a - b).max()))
    n_checked = 0
    rows_eval = [1]
    for i in range(int(z25["n_cases"])):
        z, hz = to_t(z25[f"c{i}/z"]), to_t(z25[f"c{i}/hz"])
        n = z.shape[1]
        for mode in ("r2", "adjusted_r2", "pearson", "spearman"):
            for split in (False, True):
                rows_eval[0] = z.shape[0] - z.shape[0] // 2 if split else z.shape[0]
                k = f"c{i}/lin/{mode}/{int(split)}"
                (score, corr), (z2, pred) = du.linear_disentanglement(z, hz, mode=mode, train_test_split=split)
                close(score, z25[k + "/score"], k + "/score")
                assert (corr is None) == (k + "/corr" not in z25.files)
                if corr is not None:
                    close(corr, z25[k + "/corr"], k + "/corr")
                close(pred[:16].cpu().numpy(), z25[k + "/pred_head"], k + "/pred_head")
                assert z2.shape == pred.shape
                n_checked += 1
        for mode in ("pearson", "spearman"):
            for resc in (True, False):
                k = f"c{i}/munkres/{mode}/{int(resc)}"
                (score, corr), thz = du.permutation_disentanglement(z, hz, mode=mode, solver="munkres", rescaling=resc)
                close(score, z25[k + "/score"], k + "/score"); close(corr, z25[k + "/corr"], k + "/corr")
                close(thz[:16].cpu().numpy(), z25[k + "/thz_head"], k + "/thz_head")
                n_checked += 1
        if n <= 4:
            for mode in ("r2", "pearson", "spearman"):
                for resc in (True, False):
                    for flips in (True, False):
                        k = f"c{i}/naive/{mode}/{int(resc)}/{int(flips)}"
                        (score, corr), thz = du.permutation_disentanglement(z, hz, mode=mode, solver="naive", rescaling=resc, sign_flips=flips,
                                                                            cache_permutations=(flips and resc))
                        close(score, z25[k + "/score"], k + "/score")
                        if corr is not None:
                            close(corr, z25[k + "/corr"], k + "/corr")
                        close(thz[:16].cpu().numpy(), z25[k + "/thz_head"], k + "/thz_head")
                        n_checked += 1
    return n_checked


def test_metric_modes_goldens(golden, monkeypatch):
    """All four modes x both solvers x rescaling / sign flips / train-test split (G25) through the host logic."""
    from cl_ica_amd import disentanglement_utils as du
    _host_moment_stand_ins(monkeypatch)
    assert check_metric_modes(du, golden("g25_metrics_modes.npz").z, torch.tensor) >= 60
    with pytest.raises(AssertionError):
        du.permutation_disentanglement(torch.rand(8, 2), torch.rand(8, 2), mode="r2", solver="munkres")


def test_train_mlp_cli_surface():
    """Same flags and defaults as main_mlp.py:21-127."""
    from cl_ica_amd import train_mlp
    a = train_mlp.parse_args([])
    assert (a.n, a.p, a.batch_size, a.lr, a.tau, a.c_param, a.m_param, a.c_p, a.m_p) == (10, 2, 6144, 1e-4, 1.0, 0.05, 1.0, 2, 0)
    assert (a.space_type, a.n_mixing_layer, a.n_log_steps, a.n_steps, a.more_unsupervised, a.num_eval_batches) == ("box", 3, 250, 100001, 3, 10)
    assert (a.box_min, a.box_max, a.sphere_r, a.act_fct, a.save_dir, a.seed) == (0.0, 1.0, 1.0, "leaky_relu", "", None)
    assert not (a.sphere_norm or a.box_norm or a.only_supervised or a.only_unsupervised or a.resume_training)
    s = train_mlp.sampler_spec(train_mlp.parse_args(["--space-type", "sphere", "--c-p", "0", "--m-p", "1"]), 0)
    assert (s.space, s.conditional, s.marginal) == ("sphere", "vmf", "laplace")
    s = train_mlp.sampler_spec(train_mlp.parse_args(["--space-type", "unbounded", "--c-p", "3", "--m-p", "2"]), 0)
    assert (s.space, s.conditional, s.marginal, s.c_p) == ("real", "gennorm", "normal", 3.0)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_n3_our_checkpoints_load_into_the_reference():
    """The reverse direction of tests/test_gpu_next_rows.py::test_n3_*: state dicts saved from cl_ica_amd's get_mlp /
    MixingMLP / BetaVAE_H load into the REFERENCE's modules with strict key matching (same keys, shapes, values)."""
    import subprocess, sys, tempfile, textwrap
    code = textwrap.dedent('''
        import sys, io, contextlib, warnings
        warnings.filterwarnings("ignore")
        import numpy as np, torch
        sys.path.insert(0, "%s")
        from cl_ica_amd import encoders as E, invertible_network_utils as I
        from cl_ica_amd.kitti_masks.model import BetaVAE_H as K
        ours = {}
        for head in (None, "learnable_sphere", "learnable_box", "fixed_sphere", "fixed_box"):
            ours["f_%%s" %% head] = E.get_mlp(4, 4, [12, 20, 12], output_normalization=head).state_dict()
        np.random.seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            ours["g"] = I.construct_invertible_mlp(n=4, n_layers=3, cond_thresh_ratio=0.0, n_iter_cond_thresh=50).state_dict()
        ours["kitti"] = K(z_dim=5, nc=1, box_norm=True).state_dict()
        torch.save(ours, sys.argv[1])
        for m in [k for k in sys.modules if k.split(".")[0] == "cl_ica_amd"]:
            del sys.modules[m]
        sys.path.insert(0, "/root/reference")
        import encoders as RE, invertible_network_utils as RI
        from kitti_masks.model import BetaVAE_H as RK
        ours = torch.load(sys.argv[1])
        for head in (None, "learnable_sphere", "learnable_box", "fixed_sphere", "fixed_box"):
            r = RE.get_mlp(4, 4, [12, 20, 12], output_normalization=head)
            res = r.load_state_dict(ours["f_%%s" %% head], strict=True)
            assert not res.missing_keys and not res.unexpected_keys
            for k, v in r.state_dict().items():
                assert torch.equal(v, ours["f_%%s" %% head][k])
        with contextlib.redirect_stdout(io.StringIO()):
            rg = RI.construct_invertible_mlp(n=4, n_layers=3, cond_thresh_ratio=0.0, n_iter_cond_thresh=50)
        rg.load_state_dict(ours["g"], strict=True)
        RK(z_dim=5, nc=1, box_norm=True).load_state_dict(ours["kitti"], strict=True)
        print("ok")
    ''') % ROOT_DIR
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([sys.executable, "-c", code, os.path.join(d, "ours.pth")], capture_output=True, text=True,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_lazy_stacking_mechanics_on_cpu():
    """cl_ica_amd/lazy.py with a stand-in compute function (no device code involved): two calls of one owner become ONE compute over
    the stacked batch with an ordinary autograd graph; a single call runs on first use; metadata needs no compute; optimizers flush
    pending calls before they write the parameters; an in-place parameter write makes a pending output unusable (error on use only);
    and the loss's roll detection reads torch.roll(z1, s, 0) off the graph."""
    import torch
    from cl_ica_amd import lazy, losses
    net = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.LeakyReLU(), torch.nn.Linear(5, 2))
    calls = []

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, x):
            def compute(xx):
                calls.append(xx.shape[0])
                return self.net(xx)
            return lazy.defer(self, x, compute, (x.shape[0], 2), list(self.parameters()))
    m = M()
    # (round 5) no GLOBAL optimizer hooks any more: the optimizer that holds the module's parameters is found from the module's first
    # deferred call and gets instance-level hooks; an unrelated optimizer is left alone
    opt = torch.optim.SGD(m.parameters(), lr=0.1)                                   # created before the loop, as main_mlp.py:312 does
    other = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
    x1, x2 = torch.randn(4, 3), torch.randn(4, 3)
    a = m(x1)
    assert opt in lazy._ATTACHED and other not in lazy._ATTACHED and lazy.find_optimizers(list(m.parameters())) == [opt]
    from torch.optim import optimizer as _topt
    assert lazy.flush_all not in getattr(_topt, "_global_optimizer_pre_hooks", {}).values()
    assert lazy.after_step not in getattr(_topt, "_global_optimizer_post_hooks", {}).values()
    assert isinstance(a, lazy.LazyOut) and a.shape == (4, 2) and a.dtype == torch.float32 and a.dim() == 2 and len(a) == 4 and calls == []
    b = m(x2)
    assert not isinstance(b, lazy.LazyOut) and calls == [8]                       # one compute over the stack
    av = lazy.plain(a)
    z3 = torch.roll(a, 1, 0)            # (round 6) deferred as well: a LazyRoll the loss recognises by its source; materialised, the graph shows the roll
    assert type(z3) is lazy.LazyRoll and z3.source is a and losses._rolled_rows_of(lazy.plain(z3), av)
    assert losses._rolled_rows_of(torch.roll(av, 1, 0), av) and not losses._rolled_rows_of(torch.roll(av, 1, 1), av)
    assert not losses._rolled_rows_of(torch.roll(b, 1, 0), av)
    ((a * b).sum() + z3.sum()).backward()
    got = [p.grad.clone() for p in m.parameters()]
    for p in m.parameters():
        p.grad = None
    ((net(x1) * net(x2)).sum() + torch.roll(net(x1), 1, 0).sum()).backward()
    assert all(torch.allclose(u, p.grad, atol=1e-6) for u, p in zip(got, m.parameters())) and torch.allclose(av, net(x1))
    calls.clear()
    c = m(x1)
    assert calls == [] and abs(float(c.detach().sum()) - float(net(x1).sum())) < 1e-6 and calls == [4]      # first use computes it alone
    calls.clear()
    d = m(x1)
    ref = net(x1).detach().clone()
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()                                                                      # its step pre-hook: flushed with the OLD parameters
    assert calls == [4] and torch.allclose(d.detach(), ref) and not torch.allclose(net(x1), ref)
    e = m(x1)
    with torch.no_grad():
        next(m.parameters()).add_(1.0)
    f2 = m(x2)                                                                      # the stale call is dropped, not stacked
    with pytest.raises(RuntimeError, match="modified in place"):
        e.sum()
    assert torch.allclose(lazy.plain(f2), net(x2))
    g1 = m(x1)
    with torch.no_grad():
        _ = g1 * 1
    assert lazy.plain(g1).grad_fn is not None                                       # first use inside no_grad keeps the graph
    leaf = torch.randn(5, 2, requires_grad=True)
    assert losses._rolled_rows_of(torch.roll(leaf, 2, 0), leaf) and not losses._rolled_rows_of(torch.roll(leaf * 1, 1, 0), leaf)


def test_lazy_finds_an_optimizer_built_later_on_cpu():
    """the optimizer search used to stop for good after three deferred calls -- an optimizer built AFTER a few
    evaluation calls (or re-created after a learning-rate change) never got the flush / after-step hooks, and the first pending output
    that crossed its step raised 'stale' on use.  Now the search is re-armed when the parameters' versions move without a hooked step
    (and when a pending output goes stale): the late optimizer is found at the next deferred call and its steps flush first."""
    import torch
    from cl_ica_amd import lazy
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.LeakyReLU(), torch.nn.Linear(4, 2))

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, x):
            return lazy.defer(self, x, lambda xx: self.net(xx), (x.shape[0], 2), list(self.parameters()))
    m = M()
    x = torch.randn(4, 3)
    for _ in range(lazy._SEARCH_TRIES + 2):            # evaluation calls before any optimizer exists: the search runs dry
        float(m(x).detach().sum())
    assert m._clica_opt_search[0] >= lazy._SEARCH_TRIES and not m._clica_opt_search[1]
    opt = torch.optim.SGD(m.parameters(), lr=0.1)       # built later
    assert opt not in lazy._ATTACHED
    for q in m.parameters():
        q.grad = torch.ones_like(q)
    opt.step()                                          # an un-hooked step: the versions move, HOOK_EPOCH does not
    a = m(x)                                            # ... which re-arms the search: found now
    assert opt in lazy._ATTACHED and m._clica_opt_search[1]
    ref = net(x).detach().clone()
    opt.step()                                          # hooked: the pending call is flushed with the OLD parameters first
    assert torch.allclose(lazy.plain(a).detach(), ref) and not torch.allclose(net(x), ref)
    # a re-created optimizer (learning-rate change): the old hooks die with the old object, the new one is picked up the same way
    del opt
    opt2 = torch.optim.SGD(m.parameters(), lr=0.01)
    opt2.step()
    b = m(x)
    assert opt2 in lazy._ATTACHED
    ref = net(x).detach().clone()
    opt2.step()
    assert torch.allclose(lazy.plain(b).detach(), ref)


def test_lazy_compute_many_after_step_and_shared_items_on_cpu():
    """Round-4 additions around the drop-in loop, device-free parts: `compute_many` (the stacked call as one multi-output node; None
    falls back to cat + slices), the optimizer post-step hook list, the adjacency test that lets the encoder's backward take the loss
    gradients without a copy, and the shared `.item()` of the loss scalars (blocking variant: no GPU here)."""
    from cl_ica_amd import encoders, lazy, losses
    w = torch.randn(3, 2, requires_grad=True)
    seen = []

    class Two(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w, *xs):
            ctx.save_for_backward(w, *xs)
            y = torch.cat(xs) @ w
            st, n, off, out = y.untyped_storage(), y.shape[1], 0, []
            for x in xs:
                out.append(torch.empty(0).set_(st, off * n, (x.shape[0], n), (n, 1)))
                off += x.shape[0]
            return tuple(out)

        @staticmethod
        def backward(ctx, *gs):
            w, *xs = ctx.saved_tensors
            seen.append(encoders._adjacent_rows(gs, [x.shape[0] for x in xs], w.shape[1]) is not None)
            return sum(x.t() @ g for x, g in zip(xs, gs)), *[None] * len(xs)

    class Owner:
        pass
    o = Owner()
    many = lambda xs: list(Two.apply(w, *xs))         # noqa: E731
    x1, x2 = torch.randn(4, 3), torch.randn(5, 3)
    a = lazy.defer(o, x1, lambda x: x @ w, (4, 2), [w], compute_many=many)
    b = lazy.defer(o, x2, lambda x: x @ w, (5, 2), [w], compute_many=many)
    pa = lazy.plain(a)
    assert pa.grad_fn is b.grad_fn and (pa.output_nr, b.output_nr) == (0, 1)            # ONE node, two outputs
    assert torch.allclose(pa, x1 @ w) and torch.allclose(b, x2 @ w)
    g = torch.randn(9, 2)
    torch.autograd.backward([pa, b], [g[:4], g[4:]])                                    # adjacent row blocks of one buffer ...
    torch.testing.assert_close(w.grad, torch.cat([x1, x2]).t() @ g)
    w.grad = None
    a = lazy.defer(o, x1, lambda x: x @ w, (4, 2), [w], compute_many=many); b = lazy.defer(o, x2, lambda x: x @ w, (5, 2), [w], compute_many=many)
    torch.autograd.backward([lazy.plain(a), b], [g[:4].clone(), g[4:].clone()])         # ... and separate ones
    assert seen == [True, False]
    # compute_many may decline
    a = lazy.defer(o, x1, lambda x: x @ w, (4, 2), [w], compute_many=lambda xs: None); b = lazy.defer(o, x2, lambda x: x @ w, (5, 2), [w], compute_many=lambda xs: None)
    assert torch.allclose(lazy.plain(a), x1 @ w) and b.grad_fn.name() == "SliceBackward0"
    # post-step hooks: an attached optimizer calls ours, an optimizer nobody attached does not (no global hook)
    hits = []
    lazy.AFTER_STEP.append(lambda: hits.append(1))
    try:
        opt = torch.optim.SGD([w], lr=0.1)
        w.grad = torch.zeros_like(w)
        opt.step()
        assert hits == []
        assert lazy.attach(opt) and not lazy.attach(opt)
        opt.step()
    finally:
        lazy.AFTER_STEP.pop()
    assert hits == [1] and encoders._after_step in lazy.AFTER_STEP
    # shared scalars
    src = torch.tensor([1.5, 2.5, 3.5])
    outs = src.unbind(0)
    losses._share_items(src, outs)
    assert [t.item() for t in outs] == [1.5, 2.5, 3.5]
    src[1] = 7.0                                        # written after the call: Tensor.item answers
    assert outs[1].item() == 7.0
    import copy, io
    assert copy.deepcopy(outs[0]).item() == 1.5         # the attribute travels with copies / pickles of the tensor as plain numbers
    buf = io.BytesIO(); torch.save(outs[2], buf); buf.seek(0)
    assert torch.load(buf, weights_only=False).item() == 3.5


def test_conv_stack_formulation_and_layouts_on_cpu():
    """What the conv kernels compute (csrc/linear.hip, conv section), restated with torch ops in fp64 on the CPU and pinned against
    nn.Conv2d's own arithmetic (/root/reference/kitti_masks/model.py:42-49: Conv2d(k = 4, stride 2, pad 1)): the padded space-to-depth
    tensor, the two-run row operand on the whole hs x ws grid, the data-gradient operand on the zero-padded dO grid, the weight-gradient
    contraction -- through the layout functions the GPU path uses (cl_ica_amd/conv.py: _wg, _wd, _wg_to_conv, _w5, _w5_back, and the
    index maps clica_conv_gather is driven with)."""
    from cl_ica_amd import conv
    g = torch.Generator().manual_seed(5)
    N, C_, Co, H = 3, 8, 12, 8
    x = torch.randn(N, C_, H, H, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, C_, 4, 4, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(Co, generator=g, dtype=torch.float64, requires_grad=True)
    ref = torch.nn.functional.conv2d(x, w, b, stride=2, padding=1)
    dref = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dref)
    ho, hs = H // 2, H // 2 + 1
    # S[n][sy][sx][(py, px, c)] = x[n][c][2 sy + py - 1][2 sx + px - 1]
    xp = torch.nn.functional.pad(x.detach().permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))            # NHWC, one zero pixel all round
    S = xp.view(N, hs, 2, hs, 2, C_).permute(0, 1, 3, 2, 4, 5).reshape(N * hs * hs, 4 * C_)
    rows = N * hs * hs
    flat = torch.cat([S.reshape(-1), torch.zeros((hs + 2) * 4 * C_, dtype=torch.float64)])
    A = torch.stack([torch.cat([flat[r * 4 * C_: r * 4 * C_ + 8 * C_], flat[(r + hs) * 4 * C_: (r + hs) * 4 * C_ + 8 * C_]]) for r in range(rows)])
    out = (A @ conv._wg(w).t() + b.detach()).view(N, hs, hs, Co)[:, :ho, :ho, :]
    assert rel_err(out.permute(0, 3, 1, 2).numpy(), ref.detach().numpy()) < 1e-13
    # weight / bias gradient: dO on the whole grid, zero on the non-output rows
    dO = torch.zeros(N, hs, hs, Co, dtype=torch.float64)
    dO[:, :ho, :ho, :] = dref.permute(0, 2, 3, 1)
    dO = dO.view(rows, Co)
    assert rel_err(conv._wg_to_conv(dO.t() @ A, Co, C_).numpy(), w.grad.numpy()) < 1e-13
    assert rel_err(dO.sum(0).numpy(), b.grad.numpy()) < 1e-13
    # data gradient: two runs of dO, (hs + 1) rows of zeros in front; result row r = gradient of S pixel r
    dflat = torch.cat([torch.zeros((hs + 1) * Co, dtype=torch.float64), dO.reshape(-1)])
    Ad = torch.stack([torch.cat([dflat[r * Co: r * Co + 2 * Co], dflat[(r + hs) * Co: (r + hs) * Co + 2 * Co]]) for r in range(rows)])
    dS = (Ad @ conv._wd(w)).view(N, hs, hs, 2, 2, C_)                                             # [n][sy][sx][py][px][c]
    dxp = dS.permute(0, 1, 3, 2, 4, 5).reshape(N, 2 * hs, 2 * hs, C_)[:, 1:H + 1, 1:H + 1, :]     # pixel (2 sy + py - 1, 2 sx + px - 1)
    assert rel_err(dxp.permute(0, 3, 1, 2).numpy(), x.grad.numpy()) < 1e-13
    # the 4 x 4 stage as a Linear over the 5 x 5 row grid of the stage in front (zero columns for its non-output rows)
    w5 = torch.randn(7, 6, 4, 4, generator=g, dtype=torch.float64)
    a4 = torch.randn(2, 5, 5, 6, generator=g, dtype=torch.float64)                               # grid rows incl. garbage at y = 4 / x = 4
    got = a4.reshape(2, -1) @ conv._w5(w5).t()
    want = torch.nn.functional.conv2d(a4[:, :4, :4, :].permute(0, 3, 1, 2), w5).flatten(1)
    assert rel_err(got.numpy(), want.numpy()) < 1e-13
    d5 = torch.randn(7, 5 * 5 * 6, generator=g, dtype=torch.float64)
    assert torch.equal(conv._w5_back(d5), d5.view(7, 5, 5, 6)[:, :4, :4, :].permute(0, 3, 1, 2))
    # index maps = the layout functions (what clica_conv_gather executes on the device)
    m = conv._maps(1, torch.device("cpu"))
    ws = [torch.randn(sh, generator=g) for sh in m["shapes"]]
    exp = [ws[0].permute(0, 2, 3, 1).reshape(32, 16)] + [conv._wg(ws[l]) for l in (1, 2, 3)] + [conv._wd(ws[l]) for l in (1, 2, 3)] + [conv._w5(ws[4])]
    for i, (mp, src) in enumerate(zip(m["pack"], m["pack_src"])):
        flat_w = ws[src].reshape(-1)
        got = torch.where(mp >= 0, flat_w[mp.clamp(min=0).long()], torch.zeros(()))
        assert torch.equal(got.view(exp[i].shape), exp[i]), i
    gs = [torch.randn(32, 16, generator=g), torch.randn(32, 512, generator=g), torch.randn(64, 512, generator=g), torch.randn(64, 1024, generator=g),
          torch.randn(256, 1600, generator=g)]
    back = [gs[0].view(32, 4, 4, 1).permute(0, 3, 1, 2)] + [conv._wg_to_conv(gs[l], *m["shapes"][l][:2]) for l in (1, 2, 3)] + [conv._w5_back(gs[4])]
    for i, mp in enumerate(m["unpack"]):
        assert torch.equal(gs[i].reshape(-1)[mp.long()].view(m["shapes"][i]), back[i].contiguous()), i


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` without a torchrun environment starts its own two ranks (the driver's command
    shape for N > 1 must not need a launcher).  `--launch-check` swaps the engine for a stub so the control flow -- rendezvous on
    127.0.0.1, barrier-bracketed windows, max over ranks, teardown, ONE JSON line from rank 0 -- runs on CPU ranks over gloo."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--windows", "2",
                        "--n", "10", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["launch_check"]["ranks_seen"] == 2 and line["launch_check"]["windows"] == 2
    # a mismatch between --gpus and an inherited WORLD_SIZE is still refused loudly
    r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_lazy_roll_of_a_deferred_output_on_cpu():
    """`torch.roll(z1_rec, 1, 0)` of a deferred encoder output (the reference's z3_rec, main_mlp.py:272) is itself deferred: a LazyRoll
    that remembers its source (LpSimCLRLoss recognises it and never computes it) and gives every other consumer the real rolled tensor,
    with ordinary autograd behind it; other shifts / dims / a second materialisation behave like torch.roll."""
    from cl_ica_amd import lazy
    w = torch.randn(3, 2, requires_grad=True)

    class Owner:
        pass
    o = Owner()
    x1, x2 = torch.randn(4, 3), torch.randn(4, 3)
    a = lazy.defer(o, x1, lambda x: x @ w, (4, 2), [w])
    b = lazy.defer(o, x2, lambda x: x @ w, (4, 2), [w])
    assert type(a) is lazy.LazyOut and type(b) is not lazy.LazyOut
    r = torch.roll(a, 1, 0)
    assert type(r) is lazy.LazyRoll and r.source is a and r.shift == 1 and tuple(r.shape) == (4, 2) and r._value is None
    r2 = a.roll(shifts=2, dims=(0,))
    assert type(r2) is lazy.LazyRoll and r2.shift == 2
    assert type(torch.roll(a, 1, 1)) is not lazy.LazyRoll                      # another dimension: the real thing, at once
    ref = torch.roll(x1 @ w, 1, 0)
    got = lazy.plain(r)
    assert type(got) is torch.Tensor and torch.allclose(got, ref) and lazy.plain(r) is got      # computed once
    assert torch.allclose(r2 + 0.0, torch.roll(x1 @ w, 2, 0))                  # any torch function materialises it
    (got.sum() * 2.0).backward()
    torch.testing.assert_close(w.grad, (x1.t() @ torch.full((4, 2), 2.0)))


def test_roll_deferring_view_on_cpu():
    """lazy.RollDeferring (what capture_train_step hands the recorded closure): same storage as the tensor, roll along dim 0 deferred
    (LazyRoll on the PLAIN tensor: materialising it must not defer again), everything else plain results."""
    from cl_ica_amd import lazy
    t = torch.arange(12.0).reshape(4, 3)
    v = t.as_subclass(lazy.RollDeferring)
    assert v.data_ptr() == t.data_ptr() and type(v.to(torch.float32)) is lazy.RollDeferring      # a no-op .to() keeps the view
    r = torch.roll(v, 1, 0)
    assert type(r) is lazy.LazyRoll and type(r.source) is torch.Tensor and r._value is None
    assert torch.equal(lazy.plain(r), torch.roll(t, 1, 0))
    assert type(v + 1) is torch.Tensor and type(torch.roll(v, 1, 1)) is torch.Tensor and torch.equal(torch.roll(v, 1, 1), torch.roll(t, 1, 1))
    assert torch.equal(v.roll(-1, 0) * 1.0, torch.roll(t, -1, 0))
