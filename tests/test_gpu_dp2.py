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

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("encoder_arith")]   # every test in both encoder arithmetics
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("B,n,wide,head", [(512, 10, False, None), (300, 4, False, None), (384, 10, True, None),
                                            (256, 10, True, "learnable_box"), (256, 10, False, "learnable_sphere")])
def test_two_ranks_match_single_process(tmp_path, B, n, wide, head):
    """wide = 600-wide hidden layers: the per-layer GEMM path with the bucketed, two-stream overlapped all-reduce
    (GradBuckets) under world = 2; head = a learnable output head whose gradient must be all-reduced too."""
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp2_worker.py"), str(r), str(port), str(tmp_path), str(B), str(n),
                               "wide" if wide else "narrow", str(head)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    sys.path.insert(0, HERE)
    from dp2_worker import make_problem
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    f, gW, z1, z2 = make_problem(n, 2 * B, wide, head)
    ref = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=2 * B, p=2, lr=0.0, device="cuda")
    out = ref.step_injected(z1, z2).cpu().numpy()
    r0, r1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2))
    from conftest import PARITY
    means = 0.5 * (r0["means"] + r1["means"])
    fam, case = "dp2_vs_single_process", f"B={B} n={n} wide={int(wide)} head={head}"
    for k, nm in enumerate(("loss_mean", "pos_mean", "neg_mean")):
        PARITY.check(fam, case, nm, means[k], out[k])
    li = np.concatenate([r0["loss_i"], r1["loss_i"]])
    PARITY.check(fam, case, "loss_i", li, ref.loss_out[:2 * B].cpu().numpy())
    assert np.array_equal(r0["grad"], r1["grad"])                       # all-reduce: identical on both ranks (incl. the head slot)
    g_ref = 2.0 * ref.grad_arena.cpu().numpy()
    # per parameter tensor (the last bias has an exactly-zero gradient without a head: translation invariance)
    off = 0
    names = [k for k, _ in f.named_parameters()]
    lin_last = [m for m in f if isinstance(m, torch.nn.Linear)][-1]
    last_w_scale = float(2.0 * ref._gviews[id(lin_last.weight)].abs().max().item())
    for k, prm in f.named_parameters():
        sl = slice(off, off + prm.numel()); off += (prm.numel() + 3) // 4 * 4
        if head is None and k == names[-1]:
            assert np.abs(r0["grad"][sl]).max() < 1e-6
            continue
        floor = 0.0
        if k.endswith(".bias") and k.split(".")[0] == names[-2 if head else -1].split(".")[0]:
            # last Linear's bias under a head: nearly translation invariant (a ~1e-7 residue of +-1e-4 summands, the same dY
            # rows that make up the weight gradient of that layer): measured relative to that weight gradient
            floor = last_w_scale
        PARITY.check(fam + "/grad", case, k, r0["grad"][sl], g_ref[sl], floor=floor)


@pytest.mark.parametrize("p,box", [(1, 0), (2, 1)])
def test_kitti_solver_two_ranks(tmp_path, p, box):
    """BASELINE config 5 is a data-parallel config: the KITTI-masks Solver (kitti_masks/solver.py:61-74 loop body) with
    world = 2 -- conv encoder on MIOpen, Linear / Softclip / loss / flat Adam on the HIP kernels, autograd-aware all-gather
    of the first views as the negatives pool, all-reduce of the flat gradient arena -- against the single-process Solver on
    the concatenated batch: rank-mean loss == global loss, summed gradient arena == 2 x single-process gradients, identical
    replicas after the step (and only rank 0 writes log.csv / checkpoints)."""
    import socket
    sys.path.insert(0, HERE)
    from kitti_dp2_worker import kitti_batch, solver_args
    Bp = 24                                         # pairs per rank
    # One solver iteration at a rank's batch shape in THIS process first: on a fresh box it makes MIOpen build and store its kernels for
    # these convolution shapes, so that the two ranks below find them instead of building the same kernels into the same on-disk cache at
    # the same time.  (An intermittent mismatch, 2 x in ~10 long runs and never reproduced in isolation: both ranks agree with each other and
    # disagree with two independent single-process evaluations in the first convolution's bias gradient; the concurrent first-time
    # kernel build is the one thing the two ranks do that the single process does not.  On a mismatch the test reports the ranks' local
    # gradients as well.)
    from cl_ica_amd.kitti_masks.solver import Solver as _WarmSolver
    dw = tmp_path / "warm"; dw.mkdir()
    _WarmSolver(solver_args(str(dw), p, box), data_loader=[(kitti_batch(Bp), None)]).train()
    torch.cuda.synchronize()
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "kitti_dp2_worker.py"), str(r), str(port), str(tmp_path / f"r{r}"),
                               str(Bp), str(p), str(box)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for r in range(2):
        os.makedirs(tmp_path / f"r{r}", exist_ok=True)
    outs = [q.communicate(timeout=600)[0].decode() for q in procs]
    for q, o in zip(procs, outs):
        assert q.returncode == 0, o[-3000:]
    r0, r1 = (torch.load(tmp_path / f"r{r}" / f"kitti_rank{r}.pt") for r in range(2))
    assert r0["wrote_log"] and not r1["wrote_log"]
    for k in r0["init"]:
        assert torch.equal(r0["init"][k], r1["init"][k]), k            # rank 0's initial weights were broadcast
        assert torch.equal(r0["final"][k], r1["final"][k]), k          # replicas stay identical
    assert torch.equal(r0["grad"], r1["grad"])
    # single process, concatenated batch, same initial weights
    from cl_ica_amd.kitti_masks.solver import Solver
    d = tmp_path / "single"; d.mkdir()
    x = kitti_batch(2 * Bp)
    S = Solver(solver_args(str(d), p, box), data_loader=[(x, None)])
    S.net.load_state_dict(r0["init"])
    rec = []
    inner = S.loss

    def recording(*a):
        out = inner(*a)
        rec.append(out[0].item())
        return out
    S.loss = recording
    assert S.train() is False
    from conftest import PARITY
    fam, case = "kitti_dp2_vs_single_process", f"p={p} box_norm={box} pairs/rank={Bp}"
    logged = [float(open(tmp_path / "r0" / "log.csv").read().split()[2])]          # rank 0 logs the mean over the ranks
    PARITY.check(fam, case, "loss (rank mean, as logged)", logged[0], rec[0], tol=2e-5 if abs(rec[0]) < 1 else 1e-5)
    g_ref = 2.0 * S.optim.grad_arena.cpu().numpy()
    off = 0
    for name, prm in S.net.named_parameters():
        sl = slice(off, off + prm.numel()); off += (prm.numel() + 3) // 4 * 4
        got, ref = r0["grad"].numpy()[sl], g_ref[sl]
        if name == "encoder.11.bias" and not box:
            assert np.abs(got).max() < 1e-5 * max(np.abs(g_ref).max(), 1e-30) + 1e-7       # translation invariance: exact gradient 0
            continue
        try:
            PARITY.check(fam + "/grad", case, name, got, ref)
        except AssertionError as e:      # seen once in ~10 full-suite runs (never in isolation): say WHICH side moved before failing
            d2 = tmp_path / "single_again"; d2.mkdir(exist_ok=True)
            S2 = Solver(solver_args(str(d2), p, box), data_loader=[(x, None)])
            S2.net.load_state_dict(r0["init"])
            S2.train()
            ref2 = 2.0 * S2.optim.grad_arena.cpu().numpy()[sl]
            den = max(float(np.abs(ref2).max()), 1e-30)
            loc = r0["local_grad"].numpy()[sl] + r1["local_grad"].numpy()[sl]
            raise AssertionError(f"{e}; single-process reference recomputed: |ref - ref2| / max|ref2| = {np.abs(ref - ref2).max() / den:.3e}, "
                                 f"|two-rank - ref2| / max|ref2| = {np.abs(got - ref2).max() / den:.3e}, "
                                 f"|rank0 local + rank1 local - ref2| / max|ref2| = {np.abs(loc - ref2).max() / den:.3e}; "
                                 f"got {got[:6]}, local sum {loc[:6]}, ref {ref2[:6]}") from None


def test_f16x2_guard_two_ranks_take_the_same_decision(tmp_path):
    """Data parallel: the f16x2 guard's decision must be the SAME on every rank (a rank that applies an update its peers withhold ends the
    replicas' identity).  Two ranks on one GPU over gloo; only rank 1's batch outgrows its scales (x 300): both ranks withhold that step --
    parameters, moments, counter untouched, rank 0 although its own producers saw nothing wrong -- both redo it the same number of times,
    and the replicas are bit-identical afterwards."""
    import json
    from conftest import with_free_port
    box = {}

    def run(port):
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "guard_dp2_worker.py"), str(r), str(port), str(tmp_path)],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        outs = [q.communicate(timeout=600)[0].decode() for q in procs]
        box["procs"], box["outs"] = procs, outs
        return max(q.returncode for q in procs), "\n".join(outs)
    with_free_port(run)
    for q, o in zip(box["procs"], box["outs"]):
        assert q.returncode == 0, o[-3000:]
    r0, r1 = (json.load(open(tmp_path / f"guard{r}.json")) for r in range(2))
    if r0["skip"]:
        pytest.skip("the f16x2 guard belongs to the f16x2 arithmetic")
    for r in (r0, r1):
        assert r["flags0"] == 0 and r["skipped1"] == 1 and (r["flags1"] & 2) and r["untouched"], r
        assert r["steps_done"] == r["steps0"] + 1 and r["finite"] and r["changed"] and not (r["flags2"] & 4), r
    assert r1["own_poisoned"] and not r0["own_poisoned"]                     # the verdict travelled: rank 0 withheld on rank 1's word
    assert r0["replays"] == r1["replays"] and r0["skipped2"] == r1["skipped2"]
    assert np.array_equal(np.load(tmp_path / "params0.npy"), np.load(tmp_path / "params1.npy"))
