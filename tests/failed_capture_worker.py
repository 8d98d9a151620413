"""Worker of tests/test_gpu_engine.py::test_failed_capture_leaves_the_engine_usable: a step whose collectives cannot be
captured (gloo on device tensors) -- capture() must raise AND leave the process able to step eagerly with the right result."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from cl_ica_amd import encoders
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[1]}", rank=0, world_size=1)
    res = []
    for try_capture in (False, True):
        torch.manual_seed(0)
        f = encoders.get_mlp(10, 10, [100, 500, 100])
        tr = ContrastiveTrainer(f, torch.eye(10).repeat(3, 1, 1), SamplerSpec(n=10, seed=5), batch_size=1024, p=2, lr=1e-3,
                                device="cuda", process_group=dist.group.WORLD, force_collectives=True)
        if try_capture:
            try:
                tr.capture(warmup=2)
                print("CAPTURE_UNEXPECTEDLY_WORKED")
            except Exception as e:
                print("capture failed as expected:", type(e).__name__)
                tr.graph = None
                assert not torch.cuda.is_current_stream_capturing()
                with torch.cuda.stream(tr.side_stream):
                    assert not torch.cuda.is_current_stream_capturing()
        for _ in range(4):
            o = tr.step().clone()
        torch.cuda.synchronize()
        res.append((o, tr.param_arena.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    x = torch.ones(4, device="cuda") * 2          # plain torch work on the default stream still runs
    assert float(x.sum()) == 8.0
    print("FAILED_CAPTURE_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
