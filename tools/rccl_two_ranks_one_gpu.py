#!/usr/bin/env python3
"""Does RCCL accept TWO ranks on ONE GPU?  The builder's lease has one MI355X, so a real world-2 RCCL run is only
possible if the library lets two processes share a device.  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/rccl_two_ranks_one_gpu.py
Prints what happened on each rank (an all-reduce, an all-gather and a reduce-scatter of the step's sizes, or RCCL's refusal)."""
import os, sys, traceback
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    x = torch.full((854228,), float(rank + 1), device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    y = torch.empty(world * 6144 * 10, device=dev)
    dist.all_gather_into_tensor(y, torch.full((6144 * 10,), float(rank), device=dev))
    torch.cuda.synchronize()
    print(f"[rank {rank}] RCCL accepted two ranks on one GPU: all_reduce -> {float(x[0])} (expected {world * (world + 1) / 2}), "
          f"all_gather tail {float(y[-1])}", flush=True)
    dist.destroy_process_group()
except Exception as e:      # noqa: BLE001
    msg = "".join(traceback.format_exception_only(type(e), e)).strip().replace("\n", " | ")
    print(f"[rank {rank}] RCCL REFUSED two ranks on one GPU: {msg[:600]}", flush=True)
    sys.exit(0)
