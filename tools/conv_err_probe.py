"""Error of the HIP conv stack (and of nn.Conv2d in fp32 = MIOpen, as a yardstick) against fp64 for growing batches."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import test_gpu_conv as T

def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())

for images in (2048,):
    g = torch.Generator().manual_seed(7)
    field = torch.nn.functional.avg_pool2d(torch.randn(images, 1, 64, 64, generator=g), 9, 1, 4)
    x = (field > 0.05).float().to("cuda")
    dfeats = ((torch.randn(images, 256, generator=g).abs() + 0.1) / images).to("cuda")
    convs = T._convs(1)
    gf, gg = T._run_hip(x, convs, dfeats)
    rf, rg = T._reference_fp64(x, convs, dfeats)
    for m in convs:
        m.weight.grad = None; m.bias.grad = None
    h = x
    for m in convs:
        h = torch.relu(m(h))
    h = h.flatten(1); h.backward(dfeats)
    tg = []
    for m in convs:
        tg += [m.weight.grad, m.bias.grad]
    print(images, "features hip %.2e torch %.2e" % (rel(gf, rf), rel(h.detach(), rf)))
    for i in range(10):
        print("   grad %d: hip %.2e  torch-fp32 %.2e   |ref|max %.3e" % (i, rel(gg[i], rg[i]), rel(tg[i], rg[i]), float(rg[i].abs().max())))
