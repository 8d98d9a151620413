import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Lazy accessor over a golden .npz written by tests/golden/gen_goldens.py."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    @property
    def n_cases(self):
        return int(self.z["n_cases"])

    def case(self, key):
        pre = f"{key}/"
        out = {"in": {}, "out": {}, "meta": {}}
        for k in self.z.files:
            if k.startswith(pre):
                _, grp, rest = k.split("/", 2)
                out[grp][rest] = self.z[k]
        return out

    def cases(self):
        for i in range(self.n_cases):
            yield f"c{i:03d}", self.case(f"c{i:03d}")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


def formula_weights(shape, salt):
    """Same RNG-free weight formula as tests/golden/gen_goldens.py (kept in sync by test_oracle_golden)."""
    fan_in = shape[-1] if len(shape) > 1 else shape[0]
    idx = np.arange(int(np.prod(shape)), dtype=np.float64).reshape(shape)
    w = np.sin(idx * 12.9898 + salt * 78.233) * 43758.5453
    w = w - np.floor(w)
    return ((2.0 * w - 1.0) / np.sqrt(fan_in)).astype(np.float32)


def mlp_formula_params(n, hidden, head):
    """Rebuild the parameters fill_formula() wrote into the reference module (enumeration order
    of nn.Sequential.named_parameters(): 0.weight, 0.bias, 2.weight, ...)."""
    dims = [n] + list(hidden) + [n]
    Ws, bs = [], []
    k = 0
    for l in range(len(dims) - 1):
        Ws.append(formula_weights((dims[l + 1], dims[l]), k + 1)); k += 1
        bs.append(formula_weights((dims[l + 1],), k + 1)); k += 1
    head_param = None
    if head in ("learnable_sphere", "fixed_sphere"):
        head_param = np.ones(1, np.float32)
    elif head in ("learnable_box", "fixed_box"):
        head_param = np.ones(n, np.float32)
    return Ws, bs, head_param


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    den = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b))) / den
