#!/bin/bash
# Multi-rank control flow of bench.py on a ONE-GPU box: two ranks share cuda:0, collectives over gloo, eager launches
# (--no-graph: gloo collectives cannot be captured in a HIP graph; the engine's recovery from a failed capture is covered by
# tests/test_gpu_engine.py::test_failed_capture_leaves_the_engine_usable, but gloo itself does not survive collectives that
# were issued during a capture attempt, so this script does not go there).  Checks for deadlocks / rank divergence in the
# barrier + max-over-ranks timing and that only rank 0 prints the JSON line; the numbers mean nothing.
# Config 3 the same way: tools/dp_bench_smoke.sh --latent-dim 40 --space-type sphere --p 1   (torchrun claims a bare `--n`)
set -e
cd "$(dirname "$0")/.."
export PYTHONFAULTHANDLER=1 CLICA_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 10 --warmup 3 --no-graph "$@"
