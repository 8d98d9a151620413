"""Worker of tests/test_gpu_dp2.py: one rank of a 2-rank data-parallel engine step on cuda:0 (gloo collectives on CUDA
tensors, eager launches), results written to an .npz.  usage: dp2_worker.py <rank> <port> <outdir> <B> <n>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_problem(n, B_total):
    from cl_ica_amd import encoders
    torch.manual_seed(0)
    f = encoders.get_mlp(n, n, [10 * n, 50 * n, 50 * n, 10 * n]).to("cuda")
    g = torch.Generator(device="cpu").manual_seed(1)
    gW = (torch.randn(3, n, n, generator=g) / n ** 0.5).to("cuda")
    z1 = torch.rand(B_total, n, generator=g).to("cuda")
    z2 = (z1.cpu() + 0.05 * torch.randn(B_total, n, generator=g)).clamp(0, 1).to("cuda")
    return f, gW, z1, z2


def main():
    rank, port, outdir, B, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    f, gW, z1, z2 = make_problem(n, 2 * B)
    tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=0.0, device="cuda", process_group=dist.group.WORLD)
    assert tr.world == 2 and tr.dp
    out = tr.step_injected(z1[rank * B:(rank + 1) * B], z2[rank * B:(rank + 1) * B])
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), means=out.cpu().numpy(), grad=tr.grad_arena.cpu().numpy(),
             loss_i=tr.loss_out[:B].cpu().numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
