import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_loss import _train_pair, dev
rng = np.random.default_rng(6154)
B, n = 6144, 10
z = rng.random((B, n)).astype(np.float32); zt = np.clip(z + 0.05 * rng.normal(size=(B, n)), 0, 1).astype(np.float32)
a = _train_pair(dev(z), dev(zt), dev(z), None, n, 2, 1.0, 0.5)
b = _train_pair(dev(z), dev(zt), dev(z), None, n, 2, 1.0, 0.5)
zd = dev(z)
c = _train_pair(zd, dev(zt), zd, None, n, 2, 1.0, 0.5)
for nm, (x, y) in (("sep vs sep", (a, b)), ("sep vs fused", (a, c))):
    print(nm, "o equal", torch.equal(x[0], y[0]), "dz equal", torch.equal(x[1], y[1]), "nan in dz", int(torch.isnan(x[1]).sum()), int(torch.isnan(y[1]).sum()),
          "max diff", float((x[1] - y[1]).abs().max()), "max", float(x[1].abs().max()))
