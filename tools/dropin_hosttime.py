#!/usr/bin/env python3
"""Host time of the drop-in train_step by function, both threads (the autograd engine runs our backward on its own thread, which
cProfile does not see): perf_counter wrappers around the package's entry points.     python tools/dropin_hosttime.py   (GPU box)"""
import collections, contextlib, functools, io, os, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CLICA_PKG_ROOT"):      # A/B against another copy of the Python package (same HIP library)
    os.environ.setdefault("CLICA_LIB", os.path.join(ROOT, "cl_ica_amd", "lib", "libclica_hip.so"))
    ROOT = os.path.abspath(os.environ["CLICA_PKG_ROOT"])
sys.path.insert(0, ROOT)
from cl_ica_amd import encoders, invertible_network_utils as inu, lazy, losses, ops, optim, train_mlp

acc = collections.defaultdict(lambda: [0, 0.0])
depth = [0]


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or f"{getattr(obj, '__name__', obj.__class__.__name__)}.{name}"

    @functools.wraps(f)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc[label]; e[0] += 1; e[1] += time.perf_counter() - t
    setattr(obj, name, staticmethod(g) if isinstance(obj, type) and isinstance(obj.__dict__.get(name), staticmethod) else g)


for nm in ("mlp_fwd_split", "mlp_pack_split_both", "mlp_dgrad_chain_split", "mlp_wgrad_split", "mlp_planes_from_f32", "mlp_planes_alloc",
           "mlp_signmask_alloc", "linear_dgrad", "sample", "mixing_fwd", "adam_step", "mlp_wgrad_split_kind"):
    if hasattr(ops, nm):
        wrap(ops, nm)
for cls in (encoders._MLPFusedSplitFn, losses._PairLossSymFn, losses._PairLossFn):
    for nm in ("forward", "backward"):
        wrap(cls, nm, f"{cls.__name__}.{nm}")
wrap(encoders._MLPFusedSplitFn, "_packed", "_MLPFusedSplitFn._packed")
wrap(encoders, "_inplace_ok")
wrap(lazy, "defer"); wrap(lazy._Pending, "flush", "_Pending.flush")
wrap(torch, "roll"); wrap(torch, "cat")
from cl_ica_amd import _lib
_L = _lib.load()
for nm in ("clica_mlp_fwd_split", "clica_mlp_dgrad_split", "clica_mlp_wgrad_split", "clica_mlp_pack_split_both", "clica_lp_loss_fwd",
           "clica_lp_loss_bwd_sym", "clica_adam_step_tick", "clica_adam_step", "clica_sample", "clica_mixing_fwd", "clica_tick"):
    if hasattr(_L, nm):
        wrap(_L, nm, "C:" + nm)

n, B, device = 10, 6144, "cuda"
a = types.SimpleNamespace(n=n, box_min=0.0, box_max=1.0, sphere_r=1.0, m_param=1.0, m_p=0, c_param=0.05, c_p=2, space_type="box")
latent_space = train_mlp.build_latent_space(a, train_mlp.sampler_spec(a, 0))
np.random.seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    g = inu.construct_invertible_mlp(n=n, n_layers=3, act_fct="leaky_relu", cond_thresh_ratio=0.0, n_iter_cond_thresh=25000).to(device)
loss = losses.LpSimCLRLoss(p=2, tau=1.0, simclr_compatibility_mode=True)
torch.manual_seed(0)
f = encoders.get_mlp(n_in=n, n_out=n, layers=[n * 10, n * 50, n * 50, n * 50, n * 50, n * 10]).to(device)
optimizer = optim.Adam(f.parameters(), lr=1e-4)
h = lambda z: f(g(z))   # noqa: E731
seg = collections.defaultdict(float)


def train_step(data):
    t = [time.perf_counter()]
    def lap(name):
        t.append(time.perf_counter()); seg[name] += t[-1] - t[-2]
    z1, z2 = data
    z3 = torch.roll(z1, 1, 0)
    optimizer.zero_grad(); lap("roll+zero_grad")
    z1_rec = h(z1); lap("h(z1)")
    z2_rec = h(z2); lap("h(z2)")
    z3_rec = torch.roll(z1_rec, 1, 0); lap("roll(z1_rec)")
    tot, _, lv = loss(z1, z2, z3, z1_rec, z2_rec, z3_rec); lap("loss")
    tot.backward(); lap("backward")
    optimizer.step(); lap("step")
    r = tot.item(), [v.item() for v in lv]; lap("items")
    return r


def one():
    t0 = time.perf_counter()
    z = latent_space.sample_marginal(B)
    d = (z, latent_space.sample_conditional(z, B))
    seg["sampling"] += time.perf_counter() - t0
    return train_step(d)


for _ in range(20):
    one()
torch.cuda.synchronize()
acc.clear(); seg.clear()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    one()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N
print(f"{tot * 1e3:.3f} ms per step ({1 / tot:.0f} steps/s)")
print("segments (us per step):", {k: round(v / N * 1e6, 1) for k, v in seg.items()}, "sum", round(sum(seg.values()) / N * 1e6, 1))
for k, (c, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} {c / N:5.1f} calls/step {s / N * 1e6:8.1f} us/step")
