import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import Golden, mlp_formula_params
from cl_ica_amd import encoders
from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
G = Golden("g7_trainstep.npz")
def dev(a): return torch.tensor(np.asarray(a, np.float32), device="cuda")
for key, c in G.cases():
    p = int(c["meta"]["p"]); head = str(c["meta"]["head"]); head = None if head == "None" else head
    hidden = [int(h) for h in c["meta"]["hidden"]]; n = 4
    for mode in (False, True):
        f = encoders.get_mlp(n_in=n, n_out=n, layers=list(hidden), output_normalization=head)
        Ws, bs, hp = mlp_formula_params(n, hidden, head)
        for m, W, b in zip([m for m in f if isinstance(m, torch.nn.Linear)], Ws, bs):
            m.weight.data = torch.tensor(W); m.bias.data = torch.tensor(b)
        gW = dev(np.stack([c["in"][f"g{i}"] for i in range(3)]))
        tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=64, p=p, lr=float(c["meta"]["lr"]), device="cuda", split_bf16=mode)
        for s in range(3):
            out = tr.step_injected(dev(c["in"][f"z1_{s}"]), dev(c["in"][f"z2_{s}"])).cpu().numpy()
            torch.cuda.synchronize()
            ga = tr.grad_arena
            bad = [(k, int((~torch.isfinite(tr._gviews[id(prm)])).sum())) for k, prm in f.named_parameters() if not torch.isfinite(tr._gviews[id(prm)]).all()]
            print(key, "split" if mode else "fp32", "step", s, "loss", out, "ref", c["out"]["loss"][s], "nonfinite grads", bad,
                  "params finite", bool(torch.isfinite(tr.param_arena).all()), "y finite", bool(torch.isfinite(tr.y).all()))
