"""Are the full-batch conv-stack gradient differences ReLU gate ties?  (GPU box; python tools/conv_tie_probe.py [zero_mean])
Runs the 2048-mask batch of tests/test_gpu_conv.py through the HIP stack, reads the gate bits the forward pass left in the pooled
buffers, compares them with the signs of the fp64 pre-activations, and re-evaluates the fp64 reference WITH THE HIP GATES forced:
if the remaining distance is at rounding level, the difference seen by the plain comparison is the conditioning of the comparison
(a pre-activation within fp32 rounding of zero gated differently), not arithmetic error."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import rel_err
import test_gpu_conv as T
from cl_ica_amd import conv

zero_mean = len(sys.argv) > 1 and sys.argv[1] == "zero_mean"
g = torch.Generator().manual_seed(7)
field = torch.nn.functional.avg_pool2d(torch.randn(2048, 1, 64, 64, generator=g), 9, 1, 4)
x = (field > 0.05).float().to("cuda")
if zero_mean:
    dfeats = (torch.randn(2048, 256, generator=g) / 2048).to("cuda")
else:
    dfeats = ((torch.randn(2048, 256, generator=g).abs() + 0.1) / 2048).to("cuda")
convs = T._convs(1)
out = {"arith": conv.get_arith(), "kperm": os.environ.get("CLICA_CONV16_KPERM", "1"), "zero_mean": zero_mean}
conv._POOL.clear()
got_f, got_g = T._run_hip(x, convs, dfeats)
torch.cuda.synchronize()
buf = conv._POOL[(2048, 1, x.device.index)][0]
# HIP gates of the three stages that write bits: stage l (0-based) on its row grid (l = 0: 32 x 32, l >= 1: (ho + 1)^2), [row][cout / 32] words
hip_gate = []
for l in range(3):
    cout, ho = conv.STAGES[l]
    grid = ho if l == 0 else ho + 1
    w = buf.gate[l].view(2048, grid, grid, cout // 32)[:, :ho, :ho, :].to(torch.int64) & 0xFFFFFFFF
    bits = ((w.unsqueeze(-1) >> torch.arange(32, device=w.device)) & 1).reshape(2048, ho, ho, cout)      # [img][y][x][ch]
    hip_gate.append(bits.permute(0, 3, 1, 2).bool())
hip_gate.append((buf.O4.view(2048, 5, 5, 64)[:, :4, :4, :] > 0).permute(0, 3, 1, 2))
hip_gate.append((got_f > 0).view(2048, -1, 1, 1))
c64 = []
for m in convs:
    d = torch.nn.Conv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding).to("cuda", torch.float64)
    d.weight.data = m.weight.data.double(); d.bias.data = m.bias.data.double()
    c64.append(d)

def run64(force):
    for d in c64:
        d.weight.grad = None; d.bias.grad = None
    stats = []
    for i0 in range(0, 2048, 256):
        h = x[i0:i0 + 256].double()
        for l, d in enumerate(c64):
            z = d(h)
            if True:
                hg = hip_gate[l][i0:i0 + 256]
                mism = (z > 0) != hg
                if i0 == 0 or True:
                    zm = z.detach().abs()
                    stats.append((l, int(mism.sum()), float(zm[mism].max()) if mism.any() else 0.0, float(zm.max())))
                h = z * hg.double() if force else torch.relu(z)
        (h.flatten(1) * dfeats[i0:i0 + 256].double()).sum().backward()
    grads = []
    for d in c64:
        grads += [d.weight.grad.clone(), d.bias.grad.clone()]
    return grads, stats

plain, stats = run64(False)
forced, _ = run64(True)
for l in range(5):
    n = sum(s[1] for s in stats if s[0] == l); zmax = max(s[2] for s in stats if s[0] == l); allmax = max(s[3] for s in stats if s[0] == l)
    out[f"stage{l + 1}_gate_mismatches"] = n
    out[f"stage{l + 1}_largest_mismatched_preactivation_over_max"] = zmax / allmax
names = [f"stage{i // 2 + 1}." + ("weight" if i % 2 == 0 else "bias") for i in range(10)]
out["rel_err_vs_fp64_plain"] = {n: rel_err(a.cpu().numpy(), b.cpu().numpy()) for n, a, b in zip(names, got_g, plain)}
out["rel_err_vs_fp64_with_hip_gates"] = {n: rel_err(a.cpu().numpy(), b.cpu().numpy()) for n, a, b in zip(names, got_g, forced)}
print(json.dumps(out, indent=1))
