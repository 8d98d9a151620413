"""Worker of tests/test_gpu_dp2.py::test_f16x2_guard_two_ranks_take_the_same_decision: one rank of a 2-rank data-parallel engine on cuda:0
(gloo collectives, eager launches).  Rank 1 alone is handed a batch 300 x larger than the step before: its producers poison the step, the
verdict rides through the gradient all-reduce, and BOTH ranks must withhold the update and redo it.
usage: guard_dp2_worker.py <rank> <port> <outdir>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    rank, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    from dp2_worker import make_problem
    n, B = 10, 256
    f, gW, z1, z2 = make_problem(n, 2 * B)
    tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=1e-3, device="cuda", process_group=dist.group.WORLD)
    res = dict(skip=tr.s16 is None)
    if tr.s16 is not None:
        assert tr._guard_rides, "whole-stack path: the verdict rides in the first gradient bucket"
        a, b = z1[rank * B:(rank + 1) * B], z2[rank * B:(rank + 1) * B]
        for _ in range(3):
            tr.step_injected(a, b)
        torch.cuda.synchronize()
        g0 = tr.check_arith()
        steps0 = tr.steps_done
        snap = [t.clone() for t in (tr.param_arena, tr.exp_avg, tr.exp_avg_sq)]
        big = 300.0 if rank == 1 else 1.0          # only rank 1's data grows
        tr.step_injected(a * big, b * big)
        torch.cuda.synchronize()
        g1 = tr.check_arith()
        untouched = all(torch.equal(x, y) for x, y in zip(snap, (tr.param_arena, tr.exp_avg, tr.exp_avg_sq)))
        own = tr.s16.guard()["poisoned"]            # (this rank's OWN verdict of the step: rank 0's producers saw nothing wrong)
        replays = 0
        while tr.steps_done == steps0 and replays < 16:
            tr.step_injected(a * big, b * big); replays += 1
            torch.cuda.synchronize()
        g2 = tr.check_arith()
        res.update(flags0=g0["flags"], skipped1=g1["skipped"], flags1=g1["flags"], steps_after_withheld=tr.steps_done - replays * 0, steps0=steps0,
                   untouched=bool(untouched), own_poisoned=bool(own), replays=replays, steps_done=tr.steps_done, skipped2=g2["skipped"], flags2=g2["flags"],
                   finite=bool(torch.isfinite(tr.param_arena).all()), changed=bool(not torch.equal(snap[0], tr.param_arena)))
        np.save(os.path.join(outdir, f"params{rank}.npy"), tr.param_arena.cpu().numpy())
    import json
    with open(os.path.join(outdir, f"guard{rank}.json"), "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
