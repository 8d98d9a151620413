"""Worker of tests/test_gpu_dp2.py: one rank of a 2-rank data-parallel engine step on cuda:0 (gloo collectives on CUDA
tensors, eager launches), results written to an .npz.  usage: dp2_worker.py <rank> <port> <outdir> <B> <n>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_problem(n, B_total, wide=False, head=None):
    """wide: 60n-wide hidden layers (> 512 at n = 10: the engine's per-layer GEMM path, several gradient buckets)."""
    from cl_ica_amd import encoders
    torch.manual_seed(0)
    w = 60 if wide else 50
    f = encoders.get_mlp(n, n, [10 * n, w * n, w * n, 10 * n], output_normalization=head).to("cuda")
    if head is not None:      # move the head parameter off its init value so its gradient matters
        hp = f[-1].max_abs_bound if head == "learnable_box" else f[-1].r
        hp.data.mul_(1.7)
    g = torch.Generator(device="cpu").manual_seed(1)
    gW = (torch.randn(3, n, n, generator=g) / n ** 0.5).to("cuda")
    z1 = torch.rand(B_total, n, generator=g).to("cuda")
    z2 = (z1.cpu() + 0.05 * torch.randn(B_total, n, generator=g)).clamp(0, 1).to("cuda")
    return f, gW, z1, z2


def main():
    rank, port, outdir, B, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    wide = len(sys.argv) > 6 and sys.argv[6] == "wide"
    head = sys.argv[7] if len(sys.argv) > 7 and sys.argv[7] != "None" else None
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    f, gW, z1, z2 = make_problem(n, 2 * B, wide, head)
    if rank == 1:             # replicas must come out identical even if a rank was initialised differently
        with torch.no_grad():
            for prm in f.parameters():
                prm.add_(0.01)
    tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=0.0, device="cuda", process_group=dist.group.WORLD,
                            bucket_bytes=(128 << 10) if wide else (8 << 20))
    assert tr.world == 2 and tr.dp
    if wide:
        assert not tr.fused_forward and len(tr.buckets.buckets) >= 3          # bucketed, overlapped all-reduce path
    else:
        # grouped weight gradients in two halves, the first half's all-reduce under the second half's launch
        assert tr.fused_forward and tr.wgrad_halves and len(tr.buckets.buckets) == 2
    out = tr.step_injected(z1[rank * B:(rank + 1) * B], z2[rank * B:(rank + 1) * B])
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), means=out.cpu().numpy(), grad=tr.grad_arena.cpu().numpy(),
             loss_i=tr.loss_out[:B].cpu().numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
