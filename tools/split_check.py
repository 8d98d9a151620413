"""Engine in split-bf16 mode (forward, backward chain AND weight gradients on the bf16 matrix cores) against the native
fp32 engine on the same injected batch: loss triple and every parameter gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cl_ica_amd import encoders
from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec

n, B = 10, int(os.environ.get("B", "6144"))
torch.manual_seed(0)
gW = torch.randn(3, n, n, device="cuda") / n ** 0.5
z1 = torch.rand(B, n, device="cuda"); z2 = (z1 + 0.05 * torch.randn(B, n, device="cuda")).clamp(0, 1)
res = {}
for mode in (False, True):
    torch.manual_seed(1)
    f = encoders.get_mlp(n, n, [n * 10, n * 50, n * 50, n * 50, n * 50, n * 10]).to("cuda")
    tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=0.0, device="cuda", split_bf16=mode)
    print("mode", mode, "split_bf16", tr.split_bf16, "split_wgrad", getattr(tr, "split_wgrad", None))
    out = tr.step_injected(z1, z2).cpu().numpy()
    torch.cuda.synchronize()
    res[mode] = (out, tr.grad_arena.cpu().numpy().copy(), [(k, v.grad.cpu().numpy().copy()) for k, v in f.named_parameters()])
print("loss", res[False][0], res[True][0])
for (k, a), (_, b) in zip(res[False][2], res[True][2]):
    print(k, "rel err", float(np.abs(a - b).max() / max(np.abs(a).max(), 1e-30)), "max", float(np.abs(a).max()))
