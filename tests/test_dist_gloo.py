"""Data-parallel semantics on CPU with gloo, world_size 2: the sharded loss with an autograd-aware
all-gather of negatives equals the single-process loss on the concatenated batch,
and bucketed gradient all-reduce + 1/world reproduces its parameter gradients.  The loss here is the
oracle's torch port (tests may use the oracle); the collectives under test are the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cl_ica_amd.distributed import GradBuckets, gather_negatives
    from oracle import torch_port as T
    torch.manual_seed(0)                                # identical replicas
    n, B = 4, 16
    f = T.make_mlp(n, [12, 20, 12])
    torch.manual_seed(100)
    x1_all = torch.randn(world * B, n); x2_all = x1_all + 0.1 * torch.randn(world * B, n)
    x1, x2 = x1_all[rank * B:(rank + 1) * B], x2_all[rank * B:(rank + 1) * B]
    a, b = f(x1), f(x2)
    pool = gather_negatives(a)                          # (world*B, n), differentiable w.r.t. local rows
    assert pool.shape == (world * B, n)
    loss, _, _ = T.lp_simclr_loss(a, b, pool, p=2)
    loss.backward()
    # flat arena + buckets, as the engine does
    params = list(f.parameters())
    sizes = [p.numel() for p in params]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    arena = torch.cat([p.grad.reshape(-1) for p in params]).clone()
    slices = [(int(offs[i]), int(offs[i + 2])) for i in range(len(params) - 2, -1, -2)]
    gb = GradBuckets(arena, slices, world, None, bucket_bytes=256)
    for i in range(len(slices)):
        gb.layer_done(i)
    gb.wait()
    arena /= world
    lt = loss.detach().clone()
    dist.all_reduce(lt)
    if rank == 0:
        # single-process reference on the concatenated batch: z3 = roll(z1) is a permutation of all z1
        g = T.make_mlp(n, [12, 20, 12]); g.load_state_dict(f.state_dict())
        A, Bm = g(x1_all), g(x2_all)
        ref, _, _ = T.lp_simclr_loss(A, Bm, torch.roll(A, 1, 0), p=2)
        ref.backward()
        ref_flat = torch.cat([p.grad.reshape(-1) for p in g.parameters()])
        np.save(os.path.join(out_dir, "res.npy"), np.asarray([float(lt / world), float(ref),
                float((arena - ref_flat).abs().max() / ref_flat.abs().max())]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_loss_and_grads_match_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    got, ref, gerr = np.load(tmp_path / "res.npy")
    assert abs(got - ref) < 1e-6 * abs(ref)
    assert gerr < 1e-5


def _worker_plain(rank, world, port, out_dir):
    """Counter-example: a plain (non-autograd) all_gather drops the cross-rank gradient."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cl_ica_amd.distributed import gather_negatives
    torch.manual_seed(rank)
    a = torch.randn(8, 3, requires_grad=True)
    pool = gather_negatives(a)
    (pool ** 2).sum().backward()
    ok = torch.allclose(a.grad, 2 * world * a.detach())       # every rank's pool contributes 2a
    t = torch.tensor([float(ok)])
    dist.all_reduce(t)
    if rank == 0:
        np.save(os.path.join(out_dir, "ok.npy"), np.asarray([float(t)]))
    dist.destroy_process_group()


def test_gather_negatives_backward_is_reduce_scatter(tmp_path):
    port = _free_port()
    mp.spawn(_worker_plain, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    assert np.load(tmp_path / "ok.npy")[0] == WORLD


def test_symmetric_backward_identity_two_ranks():
    """The engine's data-parallel backward: with the negatives pool = all ranks' z1 rows, rank r obtains the
    COMPLETE gradient of sum_q L_q w.r.t. its own rows from one sweep with coefficient
    C (w_ij + w_ji), given only the all-gathered log-sum-exp values -- no reduce-scatter of d/dz3.
    Checked in NumPy against the oracle's generic row + column gradients on the concatenated batch."""
    from oracle import np_oracle as O
    rng = np.random.default_rng(0)
    R, B, n, tau, alpha = 2, 24, 5, 0.8, 0.4
    for p in (1, 2, 3):
        z1 = rng.normal(size=(R * B, n)); z2 = z1 + 0.1 * rng.normal(size=(R * B, n))
        # reference semantics on the global batch: z3 = roll(z1); combined gradient on z1
        ref = O.lp_simclr_loss(z1, z2, np.roll(z1, 1, 0), p=p, tau=tau, alpha=alpha, compat=True)
        want_dz1 = ref["dz1"] + np.roll(ref["dz3"], -1, 0)
        lse_all = ref["lse"]                                   # what the ranks all-gather
        for r in range(R):
            rows = slice(r * B, (r + 1) * B)
            loc = O.lp_simclr_loss(z1[rows], z2[rows], z1, p=p, tau=tau, alpha=alpha, compat=True, grad=False)
            assert np.allclose(loc["lse"], lse_all[rows])      # row statistics do not depend on the row order of the pool
            # engine convention: every rank differentiates its LOCAL mean (C = 2(1-alpha)/B) and Adam divides by R
            C = 2 * (1 - alpha) / B
            d = z1[rows][:, None, :] - z1[None, :, :]
            neg = (np.abs(d) ** p).sum(-1)
            w = np.exp(-neg / tau - lse_all[rows][:, None]) + np.exp(-neg / tau - lse_all[None, :])
            dterm = p * np.abs(d) ** (p - 1) * np.sign(d)
            dz_neg = (-(C / tau) * w)[:, :, None] * dterm
            # positive-pair part (local rows only), from the oracle with the negatives' gradient removed
            pos_only = O.lp_simclr_loss(z1[rows], z2[rows], z1, p=p, tau=tau, alpha=alpha, compat=True)
            gp = -pos_only["dz2"]
            got = gp + dz_neg.sum(1)
            # global-mean gradient = (1/R) sum_q dL_q/dz  ->  compare with R * want (want differentiates the global mean)
            assert np.abs(got - R * want_dz1[rows]).max() < 1e-10 * max(1.0, np.abs(want_dz1).max() * R), (p, r)
