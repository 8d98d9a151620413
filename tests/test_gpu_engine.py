"""Fused train step (engine) against the reference's injected-step trajectory (G7), the autograd
drop-in path, and HIP-graph replay."""
import sys

import numpy as np
import pytest
import torch

from conftest import PARITY, mlp_formula_params
from oracle import np_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("encoder_arith")]   # every test in both encoder arithmetics


def dev(a):
    return torch.tensor(np.asarray(a, np.float32), device="cuda")


def build(n, hidden, head):
    from cl_ica_amd import encoders
    f = encoders.get_mlp(n_in=n, n_out=n, layers=list(hidden), output_normalization=head)
    Ws, bs, hp = mlp_formula_params(n, hidden, head)
    for m, W, b in zip([m for m in f if isinstance(m, torch.nn.Linear)], Ws, bs):
        m.weight.data = torch.tensor(W); m.bias.data = torch.tensor(b)
    return f


def test_trainstep_goldens(golden):
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    G = golden("g7_trainstep.npz")
    for key, c in G.cases():
        p = int(c["meta"]["p"]); head = str(c["meta"]["head"]); head = None if head == "None" else head
        hidden = [int(h) for h in c["meta"]["hidden"]]; n = 4
        f = build(n, hidden, head)
        gW = dev(np.stack([c["in"][f"g{i}"] for i in range(3)]))
        tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=64, p=p, lr=float(c["meta"]["lr"]), device="cuda")
        for s in range(5):
            out = tr.step_injected(dev(c["in"][f"z1_{s}"]), dev(c["in"][f"z2_{s}"])).cpu().numpy()
            from test_gpu_configs import traj_tol
            tl, tn = traj_tol(s)
            PARITY.check("trainstep_g7", f"{key} p={p} head={head} step{s}", "loss", out[0], c["out"]["loss"][s], tol=tl, note=tn)
            lossv = abs(float(c["out"]["loss"][s]))     # the two means are summands of the loss (see test_gpu_configs)
            PARITY.check("trainstep_g7", f"{key} p={p} head={head} step{s}", "pos_mean", out[1], c["out"]["pos"][s], floor=lossv, tol=tl, note=tn)
            PARITY.check("trainstep_g7", f"{key} p={p} head={head} step{s}", "neg_mean", out[2], c["out"]["neg"][s], floor=lossv, tol=tl, note=tn)
        assert tr.steps_done == 5
        from test_gpu_configs import P1_QUANTUM, adam_masks, adam_trajectory_check
        L = len(tr.linears)
        adam_trajectory_check("trainstep_g7/adam_params", key, f, c["out"], "param5", 7, float(c["meta"]["lr"]), 5,
                              skip=f"{2 * (L - 1)}.bias" if head is None else None, masks=adam_masks(golden, "g7", key),
                              quantum=P1_QUANTUM if p == 1 else 1e-5)


def test_engine_matches_autograd_path():
    """Same batch through (a) the drop-in modules + torch autograd and (b) the fused engine."""
    from cl_ica_amd import encoders, losses, ops
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    torch.manual_seed(1)
    n, B = 10, 1024
    for head in (None, "learnable_box"):
        f = encoders.get_mlp(n, n, [100, 500, 500, 100], output_normalization=head).to("cuda")
        gW = torch.randn(3, n, n, device="cuda") / n ** 0.5
        z1 = torch.rand(B, n, device="cuda"); z2 = (z1 + 0.05 * torch.randn_like(z1)).clamp(0, 1)
        from cl_ica_amd import lazy
        a, b = f(ops.mixing_fwd(z1, gW)), f(ops.mixing_fwd(z2, gW))
        a = lazy.plain(a)
        a.retain_grad(); b.retain_grad()
        tot, _, _ = losses.LpSimCLRLoss(p=2, simclr_compatibility_mode=True)(None, None, None, a, b, torch.roll(a, 1, 0))
        tot.backward()
        ref_grads = {k: p.grad.clone() for k, p in f.named_parameters()}
        tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=0.0, device="cuda")
        out = tr.step_injected(z1, z2)
        PARITY.check("engine_vs_autograd_dropin", f"head={head}", "loss", out[0].item(), tot.item())
        last_bias = f"{2 * (len(tr.linears) - 1)}.bias"
        for k, p in f.named_parameters():
            g = tr._gviews[id(p)]
            if k == last_bias:
                # Lp distances are translation invariant: without a head the exact gradient is 0; with the sigmoid head it is a
                # ~1e-7 residue of +-1e-4 summands.  Both paths hold summation-order noise there, nothing to compare.
                assert g.abs().max().item() < (1e-8 if head is None else 1e-6)
                continue
            floor, note = 0.0, None
            if k.endswith("max_abs_bound"):
                # d loss / d bound_k = sum_i dy_ik sigmoid(x_ik).  The loss is translation invariant in y (sum_i dy_ik = 0), and this
                # RANDOM-INIT encoder's sigmoid varies by 1e-3 around 0.5: the gradient is 1e-3 of its own summand mass
                # sum_i |dy_ik| sigmoid_ik (measured: 1e-6 against 1e-3).  A cancelling sum is judged against that mass: the bound
                # below is 1e-5 x 1e-2 x mass = 1e-7 of it (two fp32 ulps of the summands).  Measured: 1.3e-8 of the mass with the
                # pair sweep on coordinate differences (whose terms cancel pairwise), 5e-8 with the p = 2 sweep on the matrix cores.
                bound = p.detach()
                mass = (a.grad.abs() * (a.detach() / bound).abs()).sum(0) + (b.grad.abs() * (b.detach() / bound).abs()).sum(0)
                floor, note = 1e-2 * float(mass.max()), "cancelling sum: judged against 1e-2 of its summand mass"
            PARITY.check("engine_vs_autograd_dropin", f"head={head}", k, g.cpu().numpy(), ref_grads[k].cpu().numpy(), floor=floor, note=note)


def test_engine_seeded_sweep_vs_autograd_path():
    """Ten seeded random trainers -- latent dimension, encoder widths (whole-tile widths 128 / 256 / 512 included), batch
    size, head, exponent, space -- through the fused engine (one step with lr = 0 on an injected batch) against the drop-in
    modules under torch autograd on the same batch: loss and every parameter gradient at 1e-5."""
    from cl_ica_amd import encoders, losses, ops
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    rng = np.random.default_rng(11)
    widths = [16, 64, 100, 128, 256, 300, 512, 600]
    for case in range(10):
        n = int(rng.choice([2, 3, 5, 10, 16]))
        hidden = [int(rng.choice(widths)) for _ in range(int(rng.integers(1, 5)))]
        B = int(rng.choice([64, 200, 1024]))
        head = [None, "learnable_sphere", "learnable_box", "fixed_sphere", "fixed_box"][int(rng.integers(5))]
        p = int(rng.choice([1, 2, 3]))
        torch.manual_seed(100 + case)
        f = encoders.get_mlp(n, n, list(hidden), output_normalization=head).to("cuda")
        gW = torch.randn(3, n, n, device="cuda") / n ** 0.5
        z1 = torch.rand(B, n, device="cuda"); z2 = (z1 + 0.05 * torch.randn_like(z1)).clamp(0, 1)
        a, b = f(ops.mixing_fwd(z1, gW)), f(ops.mixing_fwd(z2, gW))
        tot, _, _ = losses.LpSimCLRLoss(p=p, simclr_compatibility_mode=True)(None, None, None, a, b, torch.roll(a, 1, 0))
        tot.backward()
        ref = {k: q.grad.clone() for k, q in f.named_parameters()}
        tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=p, lr=0.0, device="cuda")
        out = tr.step_injected(z1, z2)
        cid = f"#{case} n={n} hidden={hidden} B={B} head={head} p={p} fused={tr.fused_forward}"
        PARITY.check("engine_sweep_vs_autograd", cid, "loss", out[0].item(), tot.item())
        gmax = max(float(v.abs().max()) for v in ref.values())
        last_bias = f"{2 * (len(tr.linears) - 1)}.bias"
        for k, q in f.named_parameters():
            g = tr._gviews[id(q)]
            if k == last_bias and (head is None or "box" in head):
                # Lp distances are translation invariant: without a head the exact gradient is 0, behind the sigmoid-shaped box head
                # it is a small residue of a column sum of cancelling terms; a backward-stable fp32 sum is accurate relative to
                # the summands, whose scale is that of the other gradients built from the same dZ
                PARITY.check("engine_sweep_vs_autograd", cid, k + " (cancelling column sum)", g.cpu().numpy(), ref[k].cpu().numpy(), floor=gmax)
                continue
            # p = 1: the two paths run different forward kernels (one 2B-row fused pass vs two B-row calls), their embeddings
            # differ in the last bits, and sign(z1_k - z3_k) of a coordinate pair closer than that flips a whole 1 / B term
            tol, note = (5e-5, "p = 1: sign ties between two fp32 forwards") if p == 1 else (1e-5, None)
            PARITY.check("engine_sweep_vs_autograd", cid, k, g.cpu().numpy(), ref[k].cpu().numpy(), floor=1e-3 * gmax, tol=tol, note=note)


def test_graph_replay_trains():
    from cl_ica_amd import encoders
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    torch.manual_seed(0)
    n, B = 10, 2048
    f = encoders.get_mlp(n, n, [100, 500, 500, 100])
    gW = torch.randn(3, n, n) / n ** 0.5
    tr = ContrastiveTrainer(f, gW, SamplerSpec(space="box", n=n, seed=3), batch_size=B, p=2, lr=1e-3, device="cuda")
    first = tr.step().clone()
    assert abs(first[0].item() - np.log(B + 1)) < 0.05        # fresh f maps everything near 0
    tr.capture(warmup=2)
    z_before = tr.z.clone()
    for _ in range(60):
        out = tr.step()
    torch.cuda.synchronize()
    assert tr.steps_done == 1 + 60          # capture() is side-effect free (warm-up is rolled back)
    assert not torch.equal(z_before, tr.z)                     # RNG advanced across replays
    assert torch.isfinite(out).all() and out[0].item() < first[0].item() - 0.05
    assert float(tr.z.min()) >= 0.0 and float(tr.z.max()) <= 1.0


def test_train_mlp_driver_short_run(tmp_path, capsys):
    """BASELINE configs[0] shape (n=10, box, p=2, B=512) for a few steps through the CLI driver:
    step-1 unsupervised loss = ln(513), loss decreases, checkpoints have the reference keys."""
    from cl_ica_amd import train_mlp
    res = train_mlp.main(["--n", "10", "--space-type", "box", "--p", "2", "--batch-size", "512", "--n-steps", "40",
                          "--more-unsupervised", "1", "--n-log-steps", "20", "--seed", "0", "--only-unsupervised",
                          "--lr", "1e-3", "--num-eval-batches", "2", "--save-dir", str(tmp_path)])
    out = capsys.readouterr().out
    assert "Id. Lin. Disentanglement" in out and "Step: 1" in out and "linear mean" in out
    assert abs(res["losses"][0] - np.log(513)) < 2e-3
    assert res["losses"][-1] < res["losses"][0] - 0.05
    sd = torch.load(tmp_path / "unsup_f.pth")
    assert list(sd.keys()) == [f"{i}.{k}" for i in range(0, 13, 2) for k in ("weight", "bias")]
    assert list(torch.load(tmp_path / "g.pth").keys()) == ["0.weight", "2.weight", "4.weight"]


def test_train_mlp_supervised_and_simclr_paths():
    from cl_ica_amd import train_mlp
    r = train_mlp.main(["--n", "4", "--space-type", "sphere", "--p", "0", "--batch-size", "256", "--n-steps", "12",
                        "--more-unsupervised", "1", "--n-log-steps", "6", "--seed", "1", "--lr", "1e-3", "--num-eval-batches", "1"])
    assert np.isfinite(r["losses"]).all() and len(r["losses"]) == 12


def test_dp_code_path_single_rank(monkeypatch):
    """world_size-1 RCCL group with the collectives forced on: all-gather / reduce-scatter / bucketed
    all-reduce must reproduce the plain single-GPU step bit-for-bit.  (The single-GPU step normally takes the n-wide layers' weight
    gradients from the backward chain's tail -- 48-row partials, another summation order than the tiny-dimension kernel the
    data-parallel path launches; for the bit-for-bit comparison it runs with that fold off.  With it on: within 1e-5 of the largest
    gradient, test_gpu_mlp.py::test_wgrad_split_adam_equals_wgrad_then_adam.)"""
    import os
    import torch.distributed as dist
    from cl_ica_amd import encoders
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    monkeypatch.setattr(ContrastiveTrainer, "chain_tail", False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        outs = []
        for force in (False, True):
            torch.manual_seed(0)
            f = encoders.get_mlp(10, 10, [100, 500, 100])
            tr = ContrastiveTrainer(f, torch.eye(10).repeat(3, 1, 1), SamplerSpec(n=10, seed=5), batch_size=1024, p=2,
                                    lr=1e-3, device="cuda", process_group=dist.group.WORLD, force_collectives=force)
            for _ in range(3):
                o = tr.step().clone()
            outs.append((o, tr.param_arena.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        # the collectives are graph-capturable: captured replay == eager on the same seed
        res = []
        for graph in (False, True):
            torch.manual_seed(0)
            f = encoders.get_mlp(10, 10, [100, 500, 100])
            tr = ContrastiveTrainer(f, torch.eye(10).repeat(3, 1, 1), SamplerSpec(n=10, seed=5), batch_size=1024, p=2,
                                    lr=1e-3, device="cuda", process_group=dist.group.WORLD, force_collectives=True)
            if graph:
                tr.capture(warmup=2)
            for _ in range(4):
                o = tr.step().clone()
            torch.cuda.synchronize()
            res.append((o, tr.param_arena.clone()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        # the headline architecture at the full batch: the grouped weight gradients go out in two halves under data parallelism and
        # each half's launch plans its own contraction splits -- its slabs must fit the workspace (they did not: 59.3 MB against the
        # 58.5 MB of the all-layer plan; found by tools/dp_bench_smoke.sh, not by the smaller two-rank tests)
        torch.manual_seed(0)
        f = encoders.get_mlp(10, 10, [100, 500, 500, 500, 500, 100])
        tr = ContrastiveTrainer(f, torch.eye(10).repeat(3, 1, 1), SamplerSpec(n=10, seed=5), batch_size=6144, p=2, lr=1e-3,
                                device="cuda", process_group=dist.group.WORLD, force_collectives=True)
        assert tr.wgrad_halves
        ref = None
        for _ in range(3):
            o = tr.step().clone()
        torch.cuda.synchronize()
        torch.manual_seed(0)
        f2 = encoders.get_mlp(10, 10, [100, 500, 500, 500, 500, 100])
        tr2 = ContrastiveTrainer(f2, torch.eye(10).repeat(3, 1, 1), SamplerSpec(n=10, seed=5), batch_size=6144, p=2, lr=1e-3, device="cuda")
        for _ in range(3):
            o2 = tr2.step().clone()
        # (a collapsed random encoder: the gradients are at rounding level, so Adam moves an element by up to +-lr per step whichever
        #  summation order produced it -- the loss is what is compared; the parameters only have to stay within 3 steps x lr)
        assert float((o - o2).abs().max()) <= 1e-5 * float(o2.abs().max())
        assert float((tr.param_arena - tr2.param_arena).abs().max()) <= 3.1e-3 and bool(torch.isfinite(tr.param_arena).all())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,space,p", [(10, "box", 2), (40, "sphere", 1)])
def test_dry_ranks_8_plan_capture_and_run(n, space, p):
    """everything the HOST side of an 8-GPU run decides, on one GPU.  `dry_ranks=8` makes this process plan and run
    what rank 0 of an 8-rank job does (engine.ContrastiveTrainer: pool of 49 152 rows, loss workspace and stream splits for that pool,
    the two-half weight-gradient launch, gradient buckets, 1/8 gradient scale), with every collective issued on a one-rank RCCL group
    and CAPTURED into the step graph; the other ranks' rows are copies of its own.  Asserts the plan (sizes derived independently
    here), that capture + replays run, and that the dry run equals `emulate_pool_ranks=8` -- the same arithmetic without collectives --
    bit for bit in its per-row losses (the step differs only in the 1/8 gradient scale).  Runs in a subprocess: it needs its own
    process group.  Headline config (n = 10) and BASELINE configs[2] (n = 40)."""
    import os
    import subprocess
    import sys
    from conftest import with_free_port
    code = r"""
import json, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, %r)
from cl_ica_amd import encoders
from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
n, space, p, B, R = %d, %r, %d, 6144, 8
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "%d"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
def mk(**kw):
    torch.manual_seed(0)
    f = encoders.get_mlp(n, n, [10 * n, 50 * n, 50 * n, 50 * n, 50 * n, 10 * n])
    return ContrastiveTrainer(f, torch.eye(n).repeat(3, 1, 1), SamplerSpec(space=space, n=n, seed=5), batch_size=B, p=p, lr=1e-4, device="cuda", **kw)
tr = mk(process_group=dist.group.WORLD, force_collectives=True, dry_ranks=R)
plan = tr.plan_summary()
tr.capture(warmup=2)
outs = [tr.step().clone() for _ in range(3)]
torch.cuda.synchronize()
li = tr.loss_out[:B].clone()
te = mk(emulate_pool_ranks=R)
for _ in range(1):
    te.step()
torch.cuda.synchronize()
tr2 = mk(process_group=dist.group.WORLD, force_collectives=True, dry_ranks=R)
tr2.step(); torch.cuda.synchronize()
print("RESULT " + json.dumps(dict(plan=plan, captured=tr.graph is not None, finite=bool(torch.isfinite(torch.stack(outs)).all()),
                                  loss=[float(o[0]) for o in outs], first_step_rows_equal_emulated=bool(torch.equal(tr2.loss_out[:B], te.loss_out[:B])),
                                  params_finite=bool(torch.isfinite(tr.param_arena).all()))))
dist.destroy_process_group()
"""
    box = {}

    def run(port):
        box["r"] = r = subprocess.run([sys.executable, "-c", code % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n, space, p, port)],
                                      capture_output=True, text=True, timeout=900)
        return r.returncode, r.stderr
    with_free_port(run)
    r = box["r"]
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    import json
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    plan = res["plan"]
    B, R = 6144, 8
    assert plan["planned_ranks"] == R and plan["pool_rows"] == R * B == 49152 and plan["grad_scale"] == 1.0 / R
    n_params = sum((a * b + 3) // 4 * 4 + (a + 3) // 4 * 4 for a, b in zip([10 * n, 50 * n, 50 * n, 50 * n, 50 * n, 10 * n, n], [n, 10 * n, 50 * n, 50 * n, 50 * n, 50 * n, 10 * n]))
    assert plan["gradient_arena_elements"] == n_params
    cov = sorted(plan["gradient_buckets"])
    # the buckets tile the arena (up to the <= 3 padding elements behind the last parameter; in the f16x2 arithmetic + the guard's verdict slot, 4 floats)
    assert cov[0][0] == 0 and n_params - 3 <= cov[-1][1] <= n_params + 4 and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))
    ag = [c for c in plan["collectives_per_step"] if c["op"] == "all_gather"]
    assert [c["gathered_bytes"] for c in ag] == [4 * B * n * R, 4 * B * R]
    if n == 10:
        assert plan["wgrad_halves"] and len(plan["gradient_buckets"]) == 2 and plan["encoder_path"] == "whole-stack"
    else:
        assert plan["encoder_path"] == "per-layer" and len(plan["gradient_buckets"]) >= 3          # 54.6 MB of gradients in 8 MB buckets
    assert res["captured"] and res["finite"] and res["params_finite"] and res["first_step_rows_equal_emulated"]
    assert abs(res["loss"][0] - float(np.log(R * B + 1))) < 0.15 * np.log(R * B + 1)               # ~ln(49 153) at initialisation


def test_failed_capture_leaves_the_engine_usable():
    """capture() of a step whose collectives cannot be captured (gloo on device tensors) must raise and leave the process able
    to run the same steps eagerly (bench.py / train_mlp fall back to eager launches when a RCCL build refuses capture).
    Runs in a subprocess: a failed capture must not leak into this session either way."""
    import os
    import subprocess
    import sys
    from conftest import with_free_port
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "failed_capture_worker.py")
    box = {}

    def run(port):
        box["r"] = r = subprocess.run([sys.executable, worker, str(port)], capture_output=True, text=True, timeout=600)
        return r.returncode, r.stderr + r.stdout
    with_free_port(run)
    r = box["r"]
    assert r.returncode == 0 and "FAILED_CAPTURE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert "capture failed as expected" in r.stdout


def test_loss_matrix_core_switch_and_spread():
    """The engine reports the spread the p = 2 matrix-core loss sweeps have seen and can be switched to the coordinate-difference sweeps
    at run time (process-wide switch; the per-step guard is tested below): path query, results of both settings on a batch inside the
    guard's limit agree at 1e-5, a captured graph is re-captured."""
    import ctypes as C
    from cl_ica_amd import _lib, encoders
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    torch.manual_seed(2)
    n, B = 10, 1024
    f = encoders.get_mlp(n, n, [100, 500, 100]).to("cuda")
    gW = torch.eye(n, device="cuda").repeat(3, 1, 1).contiguous()
    z1 = torch.rand(B, n, device="cuda"); z2 = (z1 + 0.05 * torch.randn_like(z1)).clamp(0, 1)
    tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=0.0, device="cuda")
    lib, path = _lib.load(), C.c_int32()
    _lib.check(lib.clica_lp_loss_set_matrix_cores(2), "on for every pool (the default policy keeps a local pool on the difference sweeps)")
    try:
        _lib.check(lib.clica_lp_loss_train_path(C.byref(tr.desc), C.byref(path)), "path"); assert path.value == 1
        tr.step_injected(z1, z2)
        # (the first loss call of a workspace measures the grid of the planes and falls back itself, csrc/lp_mfma.h; a repeated step must not)
        fb = tr.loss_guard()["fallback_steps"]
        tr.step_injected(z1, z2)
        assert tr.loss_guard()["fallback_steps"] == fb and tr.loss_guard()["last_spread"] < tr.loss_guard()["limit"]
        out_m = tr.loss_out[3 * B:].clone(); g_m = tr.grad_arena.clone()
        y = tr.y[:B]
        want = 1.4426950408889634 * float(((y - y[:64].mean(0)) ** 2).sum(1).max())       # origin = mean of the pool's first 64 rows
        assert abs(tr.loss_spread() - want) <= 2e-3 * want + 1e-6
        tr.capture()
        tr.set_loss_matrix_cores(False)
        _lib.check(lib.clica_lp_loss_train_path(C.byref(tr.desc), C.byref(path)), "path"); assert path.value == 0
        assert tr.graph is not None                                   # captured again, with the other sweeps
        out_v = tr.step_injected(z1, z2).clone(); g_v = tr.grad_arena.clone()
        PARITY.check("loss_matrix_core_switch", f"n={n} B={B}", "means", out_m.cpu().numpy(), out_v.cpu().numpy())
        PARITY.check("loss_matrix_core_switch", f"n={n} B={B}", "gradient arena", g_m.cpu().numpy(), g_v.cpu().numpy())
    finally:
        _lib.check(lib.clica_lp_loss_set_matrix_cores(-1), "restore")


def _oracle_check_of_loss_state(tr, fam, case):
    """loss_i, row statistics and d loss / d y of the engine's current loss buffers against the fp64 oracle on its current embeddings."""
    B = tr.B
    y = tr.y.detach().cpu().numpy().astype(np.float64)
    orc = O.lp_simclr_loss(y[:B], y[B:], y[:B], p=2, tau=tr.desc.tau, alpha=tr.desc.alpha, compat=True, grad=False)
    g1, g2 = O.lp_symmetric_row_grads(y[:B], y[B:], y[:B], orc["lse"], orc["lse"], 2, tr.desc.tau, tr.desc.alpha, local_rows=B)
    o = tr.loss_out.cpu().numpy()
    PARITY.check(fam, case, "loss_i", o[:B], orc["loss_i"])
    PARITY.check(fam, case, "lse_i", o[2 * B:3 * B].astype(np.float64) * np.log(2.0), orc["lse"])
    PARITY.check(fam, case, "d loss / d y1", tr.dy[:B].cpu().numpy(), g1)
    PARITY.check(fam, case, "d loss / d y2", tr.dy[B:].cpu().numpy(), g2)


def test_matrix_core_guard_falls_back_inside_graph_replay(request):
    """The guard is a device-side decision: a CAPTURED step graph whose embeddings spread beyond the limit between two replays (the last
    layer is scaled under the graph) runs that replay on the coordinate-difference sweeps -- no re-capture, no host round trip -- and comes
    back to the matrix cores when the spread does.  Every state is checked against the fp64 oracle."""
    from cl_ica_amd import encoders
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    from cl_ica_amd import _lib
    torch.manual_seed(4)
    n, B = 10, 1024
    f = encoders.get_mlp(n, n, [100, 500, 100]).to("cuda")
    gW = torch.eye(n, device="cuda").repeat(3, 1, 1).contiguous()
    _lib.check(_lib.load().clica_lp_loss_set_matrix_cores(2), "on for every pool (this test is about their guard; the default policy keeps a local pool off them)")
    request.addfinalizer(lambda: _lib.check(_lib.load().clica_lp_loss_set_matrix_cores(-1), "default policy"))
    tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=0.0, device="cuda")
    if tr.loss_guard()["limit"] == 0.0:
        pytest.skip("matrix-core sweeps disabled in this environment")
    last = [m for m in f if isinstance(m, torch.nn.Linear)][-1]
    tr.capture()
    graph = tr.graph
    tr.step(); torch.cuda.synchronize()
    fb0 = tr.loss_guard()["fallback_steps"]
    tr.step(); torch.cuda.synchronize()
    g0 = tr.loss_guard()
    assert g0["fallback_steps"] == fb0 and 0 < g0["last_spread"] <= g0["limit"], g0      # a settled replay runs on the matrix cores
    _oracle_check_of_loss_state(tr, "matrix_core_guard_in_graph_replay", f"replay inside the limit (M = {g0['last_spread']:.0f})")
    scale = (4.0 * g0["limit"] / g0["last_spread"]) ** 0.5         # M grows with the square of the embedding scale: 4 x the limit (not so far that rows saturate)
    with torch.no_grad():
        last.weight.mul_(scale); last.bias.mul_(scale)
    tr.calibrate_scales()       # (f16x2 arithmetic: parameters replaced from outside by a factor of hundreds -- its scales follow; eager passes, the graph stays)
    tr.step(); torch.cuda.synchronize()
    g1 = tr.loss_guard()
    assert tr.graph is graph, "no re-capture"
    assert g1["fallback_steps"] > g0["fallback_steps"] and g1["last_spread"] > 2 * g1["limit"], g1      # (the calibration passes of the f16x2 arithmetic are loss calls too)
    _oracle_check_of_loss_state(tr, "matrix_core_guard_in_graph_replay", f"replay beyond the limit (M = {g1['last_spread']:.0f}, difference sweeps)")
    tr.step(); torch.cuda.synchronize()
    fb = tr.loss_guard()["fallback_steps"]
    assert fb == g1["fallback_steps"] + 1
    with torch.no_grad():
        last.weight.div_(scale); last.bias.div_(scale)
    tr.calibrate_scales()
    tr.step(); torch.cuda.synchronize()      # (the cloud shrank 60 x: this replay re-measures the grid ...)
    fb = tr.loss_guard()["fallback_steps"]
    tr.step(); torch.cuda.synchronize()      # (... and this one is back on the matrix cores)
    g2 = tr.loss_guard()
    assert g2["fallback_steps"] == fb and g2["last_spread"] <= g2["limit"], g2
    _oracle_check_of_loss_state(tr, "matrix_core_guard_in_graph_replay", f"replay back inside the limit (M = {g2['last_spread']:.0f})")


def test_matrix_core_loss_on_training_embeddings():
    """Parity of the p = 2 matrix-core loss sweeps WHERE THE BENCH RUNS: main_mlp.py's unnormalised encoder spreads its outputs to a
    standard deviation of ~10 within a few hundred steps (M = log2(e)/tau max |y - y_0|^2 in the thousands, where a plain
    |a|^2 + |b|^2 - 2ab expansion in fp32 is off by 1e-4).  The headline trainer after 60 / 300 / 1000 of its own steps: loss, row statistics
    and d loss / d y of the DEFAULT path (matrix cores behind the device-side guard) against the coordinate-difference sweeps on the same
    embeddings AND against the fp64 oracle, 1e-5, out to 3 000 steps; the guard's state is logged with every case."""
    import ctypes as C
    import bench
    from cl_ica_amd import _lib
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    tr = bench.build_trainer(args, torch.device("cuda"), 1)
    lib, path = _lib.load(), C.c_int32()
    _lib.check(lib.clica_lp_loss_set_matrix_cores(2), "on for every pool")
    try:
        _lib.check(lib.clica_lp_loss_train_path(C.byref(tr.desc), C.byref(path)), "path"); assert path.value == 1
        B = tr.B
        for k in (60, 300, 1000, 3000):
            while tr.steps_done < k:
                tr.step()
            tr.sample(); tr.forward()
            res = {}
            for mode in (2, 0):
                _lib.check(lib.clica_lp_loss_set_matrix_cores(mode), "switch")
                tr.loss_forward_backward()
                torch.cuda.synchronize()
                res[mode] = (tr.dy.clone(), tr.loss_out.clone())
                if mode == 2:
                    gs = tr.loss_guard()
                    spread = gs["last_spread"]
                    case = (f"after {k} steps (y std {float(tr.y.std()):.2f}, M = {spread:.0f}, "
                            f"{'difference sweeps (guard)' if spread > gs['limit'] else 'matrix cores'}, {gs['fallback_steps']} fallback steps so far)")
                    _oracle_check_of_loss_state(tr, "matrix_core_loss_training_regime_vs_oracle", case)
            _lib.check(lib.clica_lp_loss_set_matrix_cores(2), "on")
            PARITY.check("matrix_core_loss_training_regime", case, "loss_i", res[2][1][:B].cpu().numpy(), res[0][1][:B].cpu().numpy(), note="HIP vs HIP")
            PARITY.check("matrix_core_loss_training_regime", case, "lse_i", res[2][1][2 * B:3 * B].cpu().numpy(), res[0][1][2 * B:3 * B].cpu().numpy())
            PARITY.check("matrix_core_loss_training_regime", case, "d loss / d y", res[2][0].cpu().numpy(), res[0][0].cpu().numpy())
    finally:
        _lib.check(lib.clica_lp_loss_set_matrix_cores(-1), "restore")


def test_backward_chain_finishes_dy_itself_equals_the_reduction_launch(monkeypatch, request):
    """N = 1 training step without the loss's closing reduction launch (clica_lp_loss_bwd_sym_train_parts -> clica_mlp_dgrad_split_tail with
    dy_parts: the backward chain's prologue sums the pair sweep's partials, leaves the forward's means and ticks the counter) == the step
    with that launch, bit for bit: reported means, dy, every parameter and the step counter after several steps, eager and in graph
    replay, inside the guard's limit and -- the last layer scaled up -- on the difference sweeps (the other split count).  Runs with the
    matrix-core sweeps switched on for the local pool (the default policy would keep every case on the difference sweeps: one split count)
    AND with the default policy."""
    from cl_ica_amd import _lib, encoders
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    request.addfinalizer(lambda: _lib.check(_lib.load().clica_lp_loss_set_matrix_cores(-1), "default policy"))

    def run(fold: bool, scale_last: float, graph: bool):
        monkeypatch.setattr(ContrastiveTrainer, "fold_dy_reduce", bool(fold))
        torch.manual_seed(7)
        n, B = 10, 1536
        f = encoders.get_mlp(n, n, [100, 500, 500, 100]).to("cuda")
        if scale_last != 1.0:
            last = [m for m in f if isinstance(m, torch.nn.Linear)][-1]
            with torch.no_grad():
                last.weight.mul_(scale_last); last.bias.mul_(scale_last)
        gW = (torch.randn(3, n, n) / n ** 0.5).to("cuda")
        tr = ContrastiveTrainer(f, gW, SamplerSpec(space="box", n=n, seed=5), batch_size=B, p=2, lr=1e-3, device="cuda")
        if graph:
            tr.capture(warmup=2)
        outs = [tr.step().clone() for _ in range(4)]
        torch.cuda.synchronize()
        took = getattr(tr, "_dy_parts_taken", None)
        return (torch.stack(outs), tr.dy.clone(), tr.param_arena.clone(), int(tr.steps_done), tr.loss_guard(), took, bool(tr.split_bf16))

    for scale_last, graph, mfma in ((1.0, False, 2), (1.0, True, 2), (300.0, False, 2), (1.0, True, -1)):
        _lib.check(_lib.load().clica_lp_loss_set_matrix_cores(mfma), "matrix-core policy")
        a = run(True, scale_last, graph)
        b = run(False, scale_last, graph)
        assert a[3] == b[3] == 4, (a[3], b[3])
        if a[6]:      # (split arithmetics: the whole-stack chain with its tail; the native-fp32 engine keeps the reduction launch)
            assert (a[5] or 0) > 0 and not b[5], (a[5], b[5])          # the first trainer took the folded path, the second did not
        for x, y, what in zip(a[:3], b[:3], ("means", "dy", "parameters")):
            assert torch.equal(x, y), (what, scale_last, graph, float((x - y).abs().max()))
        if scale_last > 1.0 and mfma == 2 and a[4]["limit"] > 0.0:
            assert a[4]["fallback_steps"] > 0, a[4]          # this case did run on the difference sweeps


@pytest.mark.gpu
@pytest.mark.parametrize("n,p,B,space", [(10, 2, 1536, "box"), (10, 1, 1000, "box"), (4, 3, 1024, "box"), (14, 2, 6144, "box"), (10, 2, 1504, "sphere")])
def test_forward_sweep_finishes_its_rows_itself_equals_the_finalize_launch(n, p, B, space, request):
    """Single-rank training forward in ONE launch (lp_finalize.h: fwd_partial_fin_k -- the last workgroup of every owner tile merges the
    tile's partials, forms loss_i / the row statistics / the positive-pair gradient and the block sums of the means) == sweep +
    fwd_finalize_k (clica_set_tuning("lp_fused_finalize", 0)), bit for bit: the reported scalars of every step, the per-row outputs, dy,
    every parameter after several steps; eager and in graph replay (the arrival counters must come back to zero by themselves), with a
    ragged last owner tile (B % 64 != 0) and for the three specialised exponents."""
    from cl_ica_amd import _lib, encoders
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    lib = _lib.load()
    request.addfinalizer(lambda: _lib.check(lib.clica_set_tuning(b"lp_fused_finalize", 1), "clica_set_tuning"))

    def run(fused: bool, graph: bool):
        _lib.check(lib.clica_set_tuning(b"lp_fused_finalize", 1 if fused else 0), "clica_set_tuning")
        torch.manual_seed(11)
        f = encoders.get_mlp(n, n, [n * 10, n * 50, n * 50, n * 10]).to("cuda")
        gW = (torch.randn(3, n, n) / n ** 0.5).to("cuda")
        tr = ContrastiveTrainer(f, gW, SamplerSpec(space=space, n=n, seed=5), batch_size=B, p=p, lr=1e-3, device="cuda")
        if graph:
            tr.capture(warmup=2)
        outs = [tr.step().clone() for _ in range(5)]
        torch.cuda.synchronize()
        return torch.stack(outs), tr.loss_out.clone(), tr.dy.clone(), tr.param_arena.clone()

    for graph in (False, True):
        a, b = run(True, graph), run(False, graph)
        for x, y, what in zip(a, b, ("scalars", "per-row outputs", "dy", "parameters")):
            assert torch.isfinite(x).all(), what
            assert torch.equal(x, y), (what, graph, float((x - y).abs().max()))


def _guard_trainer(seed=7, n=10, B=1024, hidden=(100, 500, 500, 100), lr=1e-3):
    from cl_ica_amd import encoders
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    torch.manual_seed(seed)
    f = encoders.get_mlp(n, n, list(hidden)).to("cuda")
    gW = (torch.randn(3, n, n, device="cuda") / n ** 0.5).contiguous()
    tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n, seed=3), batch_size=B, p=2, lr=lr, device="cuda")
    return f, tr


def _oracle_check_of_applied_step(tr, f, snap, steps0, fam, case):
    """The step the engine has just APPLIED, against the fp64 oracle and against its own optimizer kernel: loss and every parameter gradient
    of the batch in tr.z at the parameters of `snap` within 1e-5, and parameters / moments == clica_adam_step(snap, that gradient, t = steps0 + 1)
    bit for bit (the update was applied exactly once, with the step count of an APPLIED step)."""
    from cl_ica_amd import ops
    B = tr.B
    lin = tr.linears
    offs, o = [], 0
    for q in f.parameters():
        offs.append(o); o += (q.numel() + 3) // 4 * 4
    pa = snap[0].cpu().numpy().astype(np.float64)
    views = [pa[o:o + q.numel()].reshape(tuple(q.shape)) for o, q in zip(offs, f.parameters())]
    P = O.MLPParams([views[2 * l] for l in range(len(lin))], [views[2 * l + 1] for l in range(len(lin))])
    Ws = [w.cpu().numpy().astype(np.float64) for w in tr.gW]
    z = tr.z.cpu().numpy().astype(np.float64)
    x = np.concatenate([O.mixing_forward(Ws, z[:B], tr.g_slope), O.mixing_forward(Ws, z[B:], tr.g_slope)])
    y, cache = O.mlp_forward(P, x)
    ref = O.lp_simclr_loss(y[:B], y[B:], np.roll(y[:B], 1, 0), p=2, compat=True)
    out = tr.loss_out[3 * B:].cpu().numpy()
    PARITY.check(fam, case, "loss", out[0], ref["loss_mean"])
    gy = np.concatenate([ref["dz1"] + np.roll(ref["dz3"], -1, 0), ref["dz2"]])
    gr = O.mlp_backward(P, cache, gy)
    gmax = max(float(np.abs(g).max()) for g in gr["dW"])
    for l, m in enumerate(lin):
        PARITY.check(fam, case, f"dW{l}", tr._gviews[id(m.weight)].cpu().numpy(), gr["dW"][l], floor=1e-3 * gmax)
        if l < len(lin) - 1:      # (the last bias: an exactly-zero gradient, translation invariance)
            PARITY.check(fam, case, f"db{l}", tr._gviews[id(m.bias)].cpu().numpy(), gr["db"][l], floor=1e-3 * gmax)
    p2, m2, v2 = (t.clone() for t in snap[:3])
    cnt = torch.tensor([steps0], dtype=torch.int32, device="cuda")
    ops.adam_step(p2, tr.grad_arena, m2, v2, cnt, tr.lr, tr.betas[0], tr.betas[1], tr.eps)
    torch.cuda.synchronize()
    assert torch.equal(p2, tr.param_arena) and torch.equal(m2, tr.exp_avg) and torch.equal(v2, tr.exp_avg_sq), "the applied update is not Adam(snapshot, gradient, t)"


def test_f16x2_guard_withholds_the_step_and_redoes_it_inside_graph_replay():
    """VERDICT r5 item 2 / ADVICE r5 (medium).  The f16x2 arithmetic runs a step on the scales of the step before; when a tensor outgrows
    them by more than 64 x between two REPLAYS of the captured step graph -- here the batch: the mixing network's last layer is scaled by
    1 000 under the graph, a `step_injected` with data of another magnitude does the same -- the replay must not reach the parameters: they,
    the moments and the step / RNG counter come out bit-identical, the guard's flag and counter say so, and the following replays redo the
    SAME batch on the fresh scales until it is applied.  The applied step is then checked against the fp64 oracle (1e-5) and against the
    optimizer kernel (bit for bit).  The host does nothing in between but read."""
    f, tr = _guard_trainer()
    if tr.s16 is None or not tr.split_f16:
        pytest.skip("the f16x2 guard belongs to the f16x2 arithmetic")
    tr._watch_versions = False                  # (nothing is written through torch here; the device-side guard alone is under test)
    tr.capture()
    graph = tr.graph
    for _ in range(6):
        tr.step()
    torch.cuda.synchronize()
    g0 = tr.check_arith()
    steps0 = tr.steps_done
    assert g0["flags"] == 0 and g0["skipped"] == 0 and steps0 == 6, g0
    snap = [t.clone() for t in (tr.param_arena, tr.exp_avg, tr.exp_avg_sq)]
    tr.gW[-1].mul_(1000.0)                      # x = g(z) grows 1 000 x: beyond the 64 x headroom of last step's scale
    tr.step(); torch.cuda.synchronize()
    g1 = tr.check_arith()
    assert g1["skipped"] == 1 and g1["new_skipped"] == 1 and (g1["flags"] & 2) and not (g1["flags"] & 4), g1
    assert tr.steps_done == steps0, "a withheld step must not advance the step / RNG counter"
    for a, b in zip(snap, (tr.param_arena, tr.exp_avg, tr.exp_avg_sq)):
        assert torch.equal(a, b), "a withheld step must leave parameters and moments untouched"
    z_withheld = tr.z.clone()
    L = len(tr.linears)
    replays = 0
    while tr.steps_done == steps0 and replays < 2 * L + 6:
        tr.step(); torch.cuda.synchronize(); replays += 1
    g2 = tr.check_arith()
    assert tr.steps_done == steps0 + 1, (replays, g2)
    assert tr.graph is graph and not (g2["flags"] & 4) and g2["skipped"] == replays, (replays, g2)
    assert torch.equal(z_withheld, tr.z), "the redo must draw the batch of the withheld step"
    assert not torch.equal(snap[0], tr.param_arena)
    _oracle_check_of_applied_step(tr, f, snap, steps0, "f16x2_guard_in_graph_replay", f"batch x 1000 between replays (applied after {replays} redo replays)")
    for _ in range(3):                           # and training goes on
        tr.step()
    torch.cuda.synchronize()
    g3 = tr.check_arith()
    assert tr.steps_done == steps0 + 4 and g3["new_skipped"] == 0 and not g3["poisoned"], g3


def test_f16x2_guard_catches_parameters_replaced_without_calibration():
    """The other way in: parameters replaced from outside (a loaded checkpoint) WITHOUT calibrate_scales -- the first layer 200 x larger.
    With the engine's own watch on the parameter versions switched off, the device-side guard alone keeps the stale-scale steps away from
    the parameters inside graph replay and lets the first healthy one through (oracle / optimizer-kernel check as above); with the watch
    on (the default), step() re-measures the scales itself and nothing is withheld."""
    f, tr = _guard_trainer(seed=8)
    if tr.s16 is None or not tr.split_f16:
        pytest.skip("the f16x2 guard belongs to the f16x2 arithmetic")
    tr.capture()
    for _ in range(4):
        tr.step()
    torch.cuda.synchronize()
    lin0 = tr.linears[0]
    # (a) the default: the version watch recalibrates
    with torch.no_grad():
        lin0.weight.mul_(200.0); lin0.bias.mul_(200.0)
    steps0 = tr.steps_done
    tr.step(); torch.cuda.synchronize()
    ga = tr.check_arith()
    assert tr.steps_done == steps0 + 1 and ga["new_skipped"] == 0, ga
    # (b) the guard alone
    tr._watch_versions = False
    with torch.no_grad():
        lin0.weight.div_(200.0); lin0.bias.div_(200.0)      # shrinking is harmless for fp16 range: one step on coarse scales, no overflow
    tr.step(); torch.cuda.synchronize()
    with torch.no_grad():
        lin0.weight.mul_(200.0); lin0.bias.mul_(200.0)
    torch.cuda.synchronize()
    tr.check_arith()
    steps0 = tr.steps_done
    snap = [t.clone() for t in (tr.param_arena, tr.exp_avg, tr.exp_avg_sq)]
    L = len(tr.linears)
    replays = 0
    while tr.steps_done == steps0 and replays < 2 * L + 6:
        tr.step(); torch.cuda.synchronize(); replays += 1
        if tr.steps_done == steps0:
            for a, b in zip(snap, (tr.param_arena, tr.exp_avg, tr.exp_avg_sq)):
                assert torch.equal(a, b), "a withheld step must leave parameters and moments untouched"
    gb = tr.check_arith()
    assert tr.steps_done == steps0 + 1 and replays >= 2 and gb["new_skipped"] == replays - 1 and not (gb["flags"] & 4), (replays, gb)
    _oracle_check_of_applied_step(tr, f, snap, steps0, "f16x2_guard_in_graph_replay", f"first layer x 200 without calibration (applied after {replays - 1} redo replays)")


def test_saved_activation_decodes_with_the_scale_its_planes_were_written_on():
    """ADVICE r5 (engine.py: saved_activation): in the f16x2 arithmetic a hidden activation exists only as scaled fp16 planes; they are
    scaled by the scale in force when they were WRITTEN -- the previous step's after a full step (the update has run since), the current
    one after a bare forward() or a calibration pass.  Both decode to the fp64 forward of the same input within 1e-5."""
    f, tr = _guard_trainer(seed=9, B=512)
    if tr.s16 is None or not tr.split_f16:
        pytest.skip("plane copies on per-tensor scales belong to the f16x2 arithmetic")
    lin = tr.linears

    def check(case):
        P = O.MLPParams([m.weight.detach().cpu().numpy() for m in lin], [m.bias.detach().cpu().numpy() for m in lin])
        _, cache = O.mlp_forward(P, tr.x.cpu().numpy())
        for l in range(len(lin) - 1):
            if tr.acts_out[l] is None:       # planes only
                PARITY.check("saved_activation_f16x2", case, f"act{l}", tr.saved_activation(l).cpu().numpy(), cache["acts"][l + 1])
    tr.lr = 0.0                                   # (the parameters the planes were computed with stay the ones the oracle sees)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    check("after a full step (planes on the previous scales)")
    with torch.no_grad():
        tr.gW[-1].mul_(8.0)                       # the activations grow 8 x (inside the 64 x headroom): the two scale sets now differ by 2^3
    tr.step(); torch.cuda.synchronize()           # planes written on the OLD scales, update -> new scales
    check("after a full step across a scale change")
    tr.sample(); tr.forward(); torch.cuda.synchronize()
    check("after a bare forward (planes on the scales in force)")
