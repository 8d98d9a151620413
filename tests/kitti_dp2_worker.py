"""Worker of tests/test_gpu_dp2.py::test_kitti_solver_two_ranks: one rank of a 2-rank data-parallel KITTI-masks Solver step
on cuda:0 (gloo collectives on CUDA tensors), results written to disk.
usage: kitti_dp2_worker.py <rank> <port> <outdir> <pairs_per_rank> <p> <box_norm 0|1>"""
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kitti_batch(pairs_total, seed=0):
    """Binary masks, rows 2i / 2i+1 = the two views of pair i (kitti_masks/dataset.py:138-145 collate layout)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = (torch.rand(pairs_total, 1, 64, 64, generator=g) < 0.1).float()
    flip = (torch.rand(pairs_total, 1, 64, 64, generator=g) < 0.02).float()
    b = (a + flip).clamp(0, 1) * (1 - flip * a)          # a few pixels changed: a shifted mask would do as well
    x = torch.stack([a, b], 1).reshape(2 * pairs_total, 1, 64, 64)
    return x


def solver_args(outdir, p, box):
    return types.SimpleNamespace(ckpt_dir=outdir, output_dir=outdir, dataset="kittimasks", cuda=True, max_iter=1, z_dim=5, num_channel=1,
                                 lr=1e-4, beta1=0.9, beta2=0.999, box_norm=bool(box), ckpt_name="last", log_step=1, save_step=1, p=p)


def main():
    rank, port, outdir, Bp, p, box = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    from cl_ica_amd.kitti_masks.solver import Solver
    torch.manual_seed(100 + rank)                # every rank draws DIFFERENT initial weights: rank 0's must win (broadcast)
    x = kitti_batch(2 * Bp)
    mine = x[rank * 2 * Bp:(rank + 1) * 2 * Bp]
    S = Solver(solver_args(outdir, p, box), data_loader=[(mine, None)])
    assert S.world == 2
    init = {k: v.detach().cpu().clone() for k, v in S.net.state_dict().items()}
    local = {}
    reduce_grads = S.optim.all_reduce_grads

    def recording_reduce():          # this rank's own gradient, as it stands when the exchange starts (diagnostics of the parent test)
        local["before"] = S.optim.grad_arena.clone()
        reduce_grads()
    S.optim.all_reduce_grads = recording_reduce
    assert S.train() is False and S.global_iter == 1            # the real loop: one iteration, rank 0 writes log.csv + checkpoint
    torch.cuda.synchronize()
    torch.save(dict(init=init, final={k: v.detach().cpu() for k, v in S.net.state_dict().items()},
                    grad=S.optim.grad_arena.cpu(), local_grad=local["before"].cpu(), wrote_log=os.path.exists(os.path.join(outdir, "log.csv"))),
               os.path.join(outdir, f"kitti_rank{rank}.pt"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
