#!/usr/bin/env python3
"""Host-side profile of the drop-in train_step (bench.py: dropin_leg, flat Adam): where the 1.5 ms per step go.
    python tools/dropin_profile.py    (GPU box)"""
import cProfile, contextlib, io, os, pstats, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CLICA_PKG_ROOT"):      # A/B against another copy of the Python package (same HIP library)
    os.environ.setdefault("CLICA_LIB", os.path.join(ROOT, "cl_ica_amd", "lib", "libclica_hip.so"))
    ROOT = os.path.abspath(os.environ["CLICA_PKG_ROOT"])
sys.path.insert(0, ROOT)
from cl_ica_amd import encoders, invertible_network_utils as inu, losses, optim, train_mlp

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


def train_step(data):
    z1, z2 = data
    z3 = torch.roll(z1, 1, 0)
    optimizer.zero_grad()
    z1_rec = h(z1); z2_rec = h(z2)
    z3_rec = torch.roll(z1_rec, 1, 0)
    tot, _, lv = loss(z1, z2, z3, z1_rec, z2_rec, z3_rec)
    tot.backward()
    optimizer.step()
    return tot.item(), [v.item() for v in lv]


def one():
    z = latent_space.sample_marginal(B)
    return train_step((z, latent_space.sample_conditional(z, B)))


for _ in range(10):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    one()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per step")
# GPU-only time of the same step: no host syncs inside, events around 50 steps would still include launch gaps -> count kernels instead
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    one()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(os.environ.get("SORT", "tottime")).print_stats(45); print(s.getvalue()[:9000])
