"""The C-ABI library loads on a CPU-only host and exports exactly what include/clica.h declares
(no compute calls: those need the GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "clica.h")
LIB = os.path.join(ROOT, "cl_ica_amd", "lib", "libclica_hip.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clica_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_header_declares_functions():
    names = declared_functions()
    assert len(names) >= 20
    for must in ("clica_lp_loss_fwd", "clica_lp_loss_bwd", "clica_linear_fwd", "clica_linear_dgrad", "clica_linear_wgrad",
                 "clica_adam_step", "clica_sample", "clica_mixing_fwd", "clica_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in clica.h but not exported: {missing}"


def test_python_binding_matches_header(lib):
    from cl_ica_amd import _lib
    declared = set(declared_functions()) - {"clica_last_error"}
    bound = set(_lib.SIGNATURES)
    assert bound == declared, (sorted(bound - declared), sorted(declared - bound))
    assert _lib.load().clica_version() >= 100


def test_host_only_entry_points(lib):
    """Planning/validation entry points are pure host code: they must work (and fail loudly) without a GPU."""
    from cl_ica_amd import _lib
    L = _lib.load()
    d = _lib.LpLossDesc(B=6144, B3=6144, n=10, p=2.0, tau=1.0, alpha=0.5, compat=1, pow=1)
    fb, bb = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.clica_lp_loss_workspace_bytes(ctypes.byref(d), ctypes.byref(fb), ctypes.byref(bb)) == 0
    assert 0 < fb.value < 64 << 20 and 0 < bb.value < 256 << 20
    bad = _lib.LpLossDesc(B=4, B3=4, n=513, p=2.0, tau=1.0, alpha=0.5, compat=1, pow=1)
    assert L.clica_lp_loss_workspace_bytes(ctypes.byref(bad), ctypes.byref(fb), ctypes.byref(bb)) == -1
    assert b"n=513" in L.clica_last_error()
    frac = _lib.LpLossDesc(B=4, B3=5, n=3, p=0.5, tau=1.0, alpha=0.5, compat=1, pow=1)
    assert L.clica_lp_loss_workspace_bytes(ctypes.byref(frac), ctypes.byref(fb), ctypes.byref(bb)) == -1
    nb = ctypes.c_size_t()
    assert L.clica_linear_wgrad_workspace_bytes(12288, 500, 500, ctypes.byref(nb)) == 0 and nb.value > 0
    assert L.clica_linear_fwd(None, 0, None, 0, None, None, 0, 1, 1, 1, 0, 0.0, None) == -1     # NULL pointers rejected
    with pytest.raises(_lib.ClicaError):
        _lib.check(-1, "probe")


def test_host_planners_on_random_shapes(lib):
    """The host-side planners behind the *_workspace_bytes entry points (stream-split planner of the loss sweeps, grouped
    weight-gradient planner, nearest-neighbour planner) on 3000 seeded random shapes: they must return, succeed, and ask for a
    sane amount of memory (no division by zero, no runaway split counts).  No GPU involved."""
    import numpy as np
    from cl_ica_amd import _lib
    rng = np.random.default_rng(0)
    fb, bb, nb = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    for _ in range(1000):
        B = int(rng.choice([1, 2, 63, 64, 65, 1000, 6144, 20000]) + rng.integers(0, 3))
        B3 = int(rng.choice([1, 7, 128, 129, 6144, 49152, 200000]) + rng.integers(0, 3))
        n = int(rng.choice([1, 2, 10, 16, 17, 40, 64, 65, 128, 300, 512]))
        p = float(rng.choice([1.0, 2.0, 2.5]))
        d = _lib.LpLossDesc(B=B, B3=B3, n=n, p=p, tau=1.0, alpha=0.5, compat=1, pow=1)
        assert lib.clica_lp_loss_workspace_bytes(ctypes.byref(d), ctypes.byref(fb), ctypes.byref(bb)) == 0, (B, B3, n)
        # partials: (splits x rows) pairs forward, (splits x rows x padded n) floats backward; splits stay <= a few dozen per row tile
        assert 0 < fb.value <= 64 * (B + B3) * 8 * 64 + (1 << 20), (B, B3, n, fb.value)
        assert 0 < bb.value <= 64 * (B + B3) * (n + 3) * 4 * 64 + (1 << 20), (B, B3, n, bb.value)
        assert lib.clica_lp_loss_train_workspace_bytes(ctypes.byref(d), ctypes.byref(nb)) == 0 and nb.value > 0
    I32 = ctypes.c_int32
    for _ in range(1000):
        L = int(rng.integers(1, 8))
        dims = [int(rng.choice([1, 3, 10, 16, 17, 100, 128, 256, 500, 512, 600, 2000])) for _ in range(L + 1)]
        M = int(rng.choice([1, 47, 48, 1000, 12288, 100000]))
        N = (I32 * L)(*dims[1:]); K = (I32 * L)(*dims[:-1])
        assert lib.clica_mlp_wgrad_workspace_bytes(M, L, N, K, ctypes.byref(nb)) == 0, (M, dims)
        dense = sum(a * b + b for a, b in zip(dims[:-1], dims[1:])) * 4
        assert 0 < nb.value <= 600 * dense + (1 << 20), (M, dims, nb.value)         # slabs: a bounded number of splits per layer
    for _ in range(1000):
        Q = int(rng.choice([1, 2, 64, 1000, 5000])); N = int(rng.choice([1, 2, 100, 250000, 3000000])); n = int(rng.integers(1, 65)); k = int(rng.integers(1, 5))
        assert lib.clica_nn_search_workspace_bytes(Q, N, n, k, ctypes.byref(nb)) == 0 and 0 < nb.value <= Q * 8 * 4 * 4096 + 4096, (Q, N, n, k, nb.value)
