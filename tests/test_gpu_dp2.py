"""Two-rank data parallelism on the REAL kernels (SURVEY.md 8(e) parity oracle): two processes share cuda:0, exchange
through gloo (eager launches), and must reproduce the single-process step on the concatenated batch:
mean of the rank losses == global loss, all-reduced (summed) gradient arena == 2 x single-process gradients."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("B,n", [(512, 10), (300, 4)])
def test_two_ranks_match_single_process(tmp_path, B, n):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp2_worker.py"), str(r), str(port), str(tmp_path), str(B), str(n)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    sys.path.insert(0, HERE)
    from dp2_worker import make_problem
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    f, gW, z1, z2 = make_problem(n, 2 * B)
    ref = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=2 * B, p=2, lr=0.0, device="cuda")
    out = ref.step_injected(z1, z2).cpu().numpy()
    r0, r1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2))
    means = 0.5 * (r0["means"] + r1["means"])
    assert np.abs(means - out).max() < 1e-5 * max(1.0, np.abs(out).max()), (means, out)
    li = np.concatenate([r0["loss_i"], r1["loss_i"]])
    assert np.abs(li - ref.loss_out[:2 * B].cpu().numpy()).max() < 1e-5 * np.abs(li).max()
    assert np.array_equal(r0["grad"], r1["grad"])                       # all-reduce: identical on both ranks
    g_ref = 2.0 * ref.grad_arena.cpu().numpy()
    scale = np.abs(g_ref).max()
    assert np.abs(r0["grad"] - g_ref).max() / scale < 2e-5, np.abs(r0["grad"] - g_ref).max() / scale
