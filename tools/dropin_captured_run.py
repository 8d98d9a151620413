#!/usr/bin/env python3
"""The reference's loop (main_mlp.py:323-333) on the drop-in modules with the closure captured once (cl_ica_amd.capture_train_step):
60 steps, for a kernel trace of what still runs per step around and inside the graph.
    rocprofv3 --kernel-trace -d gpurun_out/dropin_cap -o t --output-format csv -- python tools/dropin_captured_run.py
    python tools/dropin_timeline.py gpurun_out/dropin_cap"""
import contextlib, io, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cl_ica_amd
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


def train_step(data, loss, optimizer):
    z1, z2 = data
    z3 = torch.roll(z1, 1, 0)
    optimizer.zero_grad()
    z1_rec = h(z1); z2_rec = h(z2)
    z3_rec = torch.roll(z1_rec, 1, 0)
    tot, _, lv = loss(z1, z2, z3, z1_rec, z2_rec, z3_rec)
    tot.backward()
    optimizer.step()
    return tot.item(), [v.item() for v in lv]


def sample():
    return latent_space.sample_marginal_and_conditional(B) if hasattr(latent_space, "sample_marginal_and_conditional") else None


data = sample()
if data is None:
    z = latent_space.sample_marginal(B); data = (z, latent_space.sample_conditional(z, B))
step = cl_ica_amd.capture_train_step(train_step, data, loss, optimizer)
for _ in range(60):
    d = sample()
    if d is None:
        z = latent_space.sample_marginal(B); d = (z, latent_space.sample_conditional(z, B))
    out = step(d, loss, optimizer)
torch.cuda.synchronize()
print("last", out)
