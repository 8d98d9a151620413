"""Debug: engine (symmetric loss backward) vs the autograd drop-in path vs the fp64 oracle at n=40, B=6144, p=1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import np_oracle as O
from cl_ica_amd import encoders, losses
from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
torch.manual_seed(5)
n, B = 40, int(sys.argv[1]) if len(sys.argv) > 1 else 6144
f = encoders.get_mlp(n, n, [n * 10, n * 50, n * 50, n * 50, n * 50, n * 10])
rng = np.random.default_rng(11)
gW = (rng.normal(size=(3, n, n)) / np.sqrt(n)).astype(np.float32)
z1 = rng.normal(size=(B, n)); z1 /= np.linalg.norm(z1, axis=1, keepdims=True)
z2 = z1 + 0.05 * rng.normal(size=(B, n)); z2 /= np.linalg.norm(z2, axis=1, keepdims=True)
z1 = z1.astype(np.float32); z2 = z2.astype(np.float32)
dev = lambda a: torch.tensor(a, device="cuda")
tr = ContrastiveTrainer(f, dev(gW), SamplerSpec(space="sphere", n=n), batch_size=B, p=1, lr=0.0, device="cuda")
out = tr.step_injected(dev(z1), dev(z2)).cpu().numpy()
y = tr.y.clone()
print("y stats", y.abs().max().item(), y.std(0).mean().item(), "loss", out)
a = y[:B].clone().requires_grad_(True); b = y[B:].clone().requires_grad_(True)
tot, per, _ = losses.LpSimCLRLoss(p=1, simclr_compatibility_mode=True)(None, None, None, a, b, torch.roll(a, 1, 0))
tot.backward()
ga = torch.cat([a.grad, b.grad]).cpu().numpy()
ge = tr.dy.cpu().numpy()
yn = y.cpu().numpy().astype(np.float64)
ref = O.lp_simclr_loss(yn[:B], yn[B:], np.roll(yn[:B], 1, 0), p=1, compat=True)
go = np.concatenate([ref["dz1"] + np.roll(ref["dz3"], -1, 0), ref["dz2"]])
sc = np.abs(go).max()
print("scale", sc)
for nm, g in (("engine", ge), ("autograd", ga)):
    e = np.abs(g - go)
    print(nm, "max err/scale", e.max() / sc, "rows>1e-5:", int((e.max(1) / sc > 1e-5).sum()), "argmax row", int(e.max(1).argmax()))
e = np.abs(ge - ga); print("engine vs autograd", e.max() / sc)
# exact ties?
d = yn[:B][:, None, :8] - yn[:B][None, :, :8] if B <= 1024 else None
yy = y[:B]
ties = 0
for k in range(4):
    col = yy[:, k].contiguous()
    u = torch.unique(col).numel(); ties += B - u
print("duplicate values in first 4 columns:", ties)
