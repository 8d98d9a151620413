#!/usr/bin/env python3
"""bench.py -- training steps/sec of the cl-ica contrastive hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], = SURVEY.md M1): main_mlp.py --n 10 --n-mixing-layer 3 --p 2
--batch-size 6144, box [0,1], uniform marginal, truncated normal conditional sigma=0.05, tau=1,
compat-mode LpSimCLRLoss, Adam lr=1e-4.  One "step" = one B=6144 batch through
sample -> g -> f fwd -> loss -> backward -> Adam (main_mlp.py:258-285,328) -- nothing skipped.
At N GPUs every rank processes its own B=6144 batch per global step against the all-gathered
N*B negatives pool (weak scaling); `value` counts batches/s over all ranks = N * global_steps/s.

Encoder arithmetic (round 3): the headline runs the split-bf16 mode -- fp32 EMULATION on the bf16 matrix cores (every fp32
operand split exactly into three bf16 pieces, six piece products, fp32 accumulate; measured error vs fp64 at the native fp32
kernels' level) for the forward stack, the backward data chain and the weight gradients.  `dtype` says so; the `native_fp32`
leg is the same step on the fp32-MFMA kernels (`--native-fp32` makes that the headline instead).

Besides the contract fields the JSON line carries
  roofline      -- the step's dominant kernel symbol (the whole-stack kernel `mlp_split_k` / `mlp_fwd_k`: forward
                   stack + backward data chain; per-layer `gemm_k` for wide encoders) timed with HIP events inside
                   training steps: ISSUED bf16 flops (6 x algorithmic) / time vs the 2.5 PFLOP/s dense bf16 peak, with
                   `fp32_equivalent` = algorithmic flops / time vs the 157.3 TFLOP/s fp32 matrix peak next to it
  native_fp32   -- value + roofline of the same step on the native fp32-MFMA kernels
  ranks         -- N > 1: per rank the world size it saw and the wall time of each collective of the step
  cpu_baseline  -- oracle/torch_port.py (the reference's op sequence in PyTorch CPU ops) timed on
                   this box's host cores, rank 0 at N=1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_F16_MFMA_TFLOPS = 2500.0   # same guide: dense fp16 MFMA (= the bf16 figure)
PEAK_HBM_GBS = 8000.0           # same guide: HBM3E peak (about 6.3 TB/s achievable)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA (the split-bf16 mode issues six bf16 products per fp32 product)
NOMINAL_GHZ = 2.4                 # MI355X peak engine clock (MI355X_MICROARCH.md): what the peak TFLOP/s figures are quoted at
PEAK_FP32_VALU_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; the median window is reported")
    ap.add_argument("--n", "--latent-dim", dest="n", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=6144)
    ap.add_argument("--p", type=int, default=2)
    ap.add_argument("--space-type", default="box", choices=("box", "sphere"))
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="single-stream backward (A/B switch)")
    ap.add_argument("--no-fused-forward", action="store_true", help="per-layer forward GEMMs instead of the one-launch stack (A/B switch)")
    ap.add_argument("--native-fp32", action="store_true",
                    help="headline on the native fp32-MFMA encoder kernels instead of the default split-bf16 arithmetic (exact 3-way "
                         "bf16 splits of both fp32 operands, six bf16-MFMA products, fp32 accumulate: fp32 emulation, fp32-grade error)")
    ap.add_argument("--split-bf16", action="store_true", help="(default since round 3; accepted for compatibility)")
    ap.add_argument("--no-native-leg", "--no-split-probe", dest="no_native_leg", action="store_true",
                    help="skip the `native_fp32` leg (the same step on the fp32-MFMA kernels, N = 1 only)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` leg (BASELINE config 3 on one rank: n = 40, sphere, p = 1, B = 6144, with the local pool and with "
                         "the emulated 49 152-row pool of the 8-GPU job; N = 1 only)")
    ap.add_argument("--config", default=None, choices=("c4", "c5"),
                    help="run ONLY the BASELINE configs[3] (c4: 3DIdent ResNet-18 train_step, batch 1024) or configs[4] (c5: KITTI-masks "
                         "Solver iteration, batch 2048) leg and print its JSON line (used under rocprofv3: tools/profile_round.sh)")
    ap.add_argument("--dry-ranks", type=int, default=0,
                    help="one GPU, one process: plan, capture and run the step exactly as rank 0 of an R-rank data-parallel job would "
                         "(pool of R x B rows, workspaces, weight-gradient halves, gradient buckets, every collective on a one-rank RCCL "
                         "group inside the step graph; the other ranks' rows are copies).  Prints the plan and this rank's step time -- NOT "
                         "a multi-GPU measurement")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not spawn the two short rocprofv3 --pmc passes that measure the dominant kernel's HBM-side bytes "
                         "(roofline.traffic then falls back to the committed profile)")
    ap.add_argument("--no-conv-configs", action="store_true", help="skip the c4 / c5 legs of `secondary`")
    ap.add_argument("--no-dry-leg", action="store_true", help="skip the `dry_ranks_8` leg (rank 0 of an 8-rank job planned, captured and run on this GPU; N = 1 only)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only the multi-rank control flow (rendezvous, barrier-bracketed windows, max over ranks, one JSON line from rank 0) around a "
                         "stub step; works on CPU ranks over gloo.  Nothing is measured")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in leg (the reference's train_step on the swapped-in modules, N = 1 only)")
    return ap.parse_args()


def build_trainer(args, device, world, split_bf16=None, emulate_pool_ranks=1, dry_ranks=1, split_arith=None):
    from cl_ica_amd import encoders, invertible_network_utils as inu
    from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
    import contextlib, io
    n = args.n
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        g = inu.construct_invertible_mlp(n=n, n_layers=3, act_fct="leaky_relu", cond_thresh_ratio=0.0,
                                         n_iter_cond_thresh=25000 if n <= 10 else 2000)
    torch.manual_seed(0)   # identical replicas on every rank
    f = encoders.get_mlp(n_in=n, n_out=n, layers=[n * 10, n * 50, n * 50, n * 50, n * 50, n * 10])
    spec = SamplerSpec(space=args.space_type, n=n, box=(0.0, 1.0), marginal="uniform", conditional="normal", c_param=0.05, seed=0)
    return ContrastiveTrainer(f, g.weight_stack(), spec, batch_size=args.batch_size, p=args.p, tau=1.0, lr=1e-4,
                              device=device, process_group=None if (world == 1 and dry_ranks == 1) else dist.group.WORLD,
                              overlap_backward=not args.no_overlap, fused_forward=not args.no_fused_forward,
                              split_bf16=(not args.native_fp32) if split_bf16 is None else split_bf16,
                              emulate_pool_ranks=emulate_pool_ranks, dry_ranks=dry_ranks, force_collectives=dry_ranks > 1,
                              split_arith=split_arith)


def _graph_time(fns, reps):
    """Capture the launches `fns` into a HIP graph, replay it `reps` times between two HIP events on the
    launch stream; returns seconds per replay (kernel time + ~1 us in-graph launch boundary each)."""
    for fn in fns:
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for fn in fns:
            fn()
    graph.replay(); torch.cuda.synchronize()
    # the chip needs tens of milliseconds of continuous work to settle at its sustained clock after an idle gap (graph capture,
    # host-side set-up): the rocprofv3 trace of this leg showed the first replays after a gap 10-12 % slower than the same
    # launch in the training loop, decaying over ~20 ms.  Replay untimed for >= 40 ms first.
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        for _ in range(10):
            graph.replay()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        graph.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def roofline_leg(tr, reps=20):
    """Per KERNEL SYMBOL (= one row of the rocprofv3 summary) timing of the encoder launches the step
    really issues (same ops, same buffers, same order as ContrastiveTrainer.forward/backward).  For each
    symbol the step's launches of it are captured into a HIP graph that is replayed `reps` times between
    two HIP events on the launch stream, so the figure is kernel time, not host launch latency --
    comparable with the rocprofv3 average for that symbol (profiles/).  FLOPs are algorithmic: 2*M*N*K per
    Linear application, nothing for the epilogues.
    Narrow encoders (every width <= 512, the north-star n=10 config) run the whole forward stack and the
    whole backward data chain as ONE `mlp_fwd_k` launch each; that symbol is then the step's dominant
    kernel (~43 % of kernel time) and is the `roofline` entry, averaged over its two launches per step
    exactly as the profiler averages them.  Wide encoders (n=40) run per-layer `gemm_k` launches and the
    entry is the forward GEMM instance with the largest share.  All symbols are listed in `kernels`."""
    from cl_ica_amd import ops
    R = 2 * tr.B
    L = len(tr.linears)
    groups = {}

    def add(key, flops, fn):
        g = groups.setdefault(key, {"fns": [], "flops": 0.0})
        g["fns"].append(fn); g["flops"] += flops

    def gemm_key(op, N, K):
        tm, tn, waves, splits = ops.linear_plan(op, R, N, K)
        vec = (N % 4 == 0 and K % 4 == 0)
        layout = {"fwd": "true, true, 0", "dgrad": "true, false, 1", "wgrad": "false, false, 2"}[op]
        if op == "wgrad" and vec and (tm, tn, waves) == (128, 128, 8):     # direct global->LDS body (one-problem group)
            return ("linear_wgrad", "clica::gemm::wgrad_group_k<128, 128, 2, 4, 3> (+ slab_reduce_k)")
        return ("linear_" + op, f"clica::gemm::gemm_k<{tm}, {tn}, ...{waves} waves..., {layout}, {'true' if vec else 'false'}>"
                + (" (+ slab_reduce_k)" if op == "wgrad" else ""))

    chain = getattr(tr, "chain", set())
    chain_key = ("linear_fwd+linear_dgrad (wide layers)", "clica::wsplit::gemm_split_k<%d>" % (1 if getattr(tr, "split_f16_wide", False) else 0))
    wide_w = bool(getattr(tr, "split_wgrad_wide", False))
    split = bool(getattr(tr, "split_bf16", False))
    fused_sym = (("clica::fmlp::mlp_split_k<1>" if getattr(tr, "split_f16", False) else "clica::fmlp::mlp_split_k<0>") if split
                 else "clica::fmlp::mlp_fwd_k<true, false>")
    fused_key = ("mlp_fwd+mlp_dgrad", fused_sym)
    if tr.fused_forward:
        fl = sum(2.0 * R * lin.out_features * lin.in_features for lin in tr.linears)

        def fwd_fn():
            tr._packed_current = True          # the fragment-order copies of the last step are still valid (lr = whatever: same layout)
            tr.forward()
        add(fused_key, fl, fwd_fn)
        add(("mlp_fwd", fused_sym + " [forward stack launch]"), fl, fwd_fn)
    else:
        cur = tr.x
        for l, lin in enumerate(tr.linears):
            N, K = lin.out_features, lin.in_features
            if l in chain:        # wide on both sides: split-bf16 GEMM with fused epilogue (forward and data gradient share the symbol)
                nxt = (l + 1) in chain
                add(chain_key, 2.0 * R * N * K,
                    lambda lin=lin, l=l, nxt=nxt: ops.linear_split_fwd(
                        tr.xT[l], tr.wT[l], lin.bias, R, lin.out_features, lin.in_features, l < L - 1, tr.slope,
                        yT=tr.xT[l + 1] if nxt else None, yN=tr.xin_planes[l + 1] if (l + 1 < L and tr.wide_kinds[l + 1] == 0) else None,
                        yN_ones=True, y=None if nxt else tr.acts[l],
                        **(dict(state=tr.s16, layer=l) if getattr(tr, "split_f16_wide", False) else {})))
            else:
                add(gemm_key("fwd", N, K), 2.0 * R * N * K,
                    lambda cur=cur, lin=lin, l=l: ops.linear_fwd(cur, lin.weight, lin.bias, leaky=(l < L - 1), slope=tr.slope, out=tr.acts[l]))
            cur = tr.acts[l]
    g_top = tr.dy if tr.head is None else tr.dpre
    if tr.fused_backward:
        chain = list(range(L - 1, 0, -1))
        fl = sum(2.0 * R * tr.linears[l].out_features * tr.linears[l].in_features for l in chain)

        def chain_fn():
            tr._packed_current = True
            tr.backward_chain(g_top)
        add(fused_key, fl, chain_fn)
        add(("mlp_dgrad", fused_sym + " [backward data chain launch]"), fl, chain_fn)
        if tr.grouped_wgrad:
            flw = sum(2.0 * R * lin.out_features * lin.in_features for lin in tr.linears)
            wsym = ("clica::wsplit::wgrad_split_k (+ wgrad_tiny_k + slab_reduce_group_k)" if getattr(tr, "split_wgrad", False)
                    else "clica::gemm::wgrad_group_k<128, 128, 2, 4, 3> (+ wgrad_tiny_k + slab_reduce_group_k)")
            add(("mlp_wgrad", wsym), flw, lambda: tr.weight_grads(g_top))
        else:
            for l in reversed(range(L)):
                lin = tr.linears[l]
                N, K = lin.out_features, lin.in_features
                gl = g_top if l == L - 1 else tr.dz[l]
                inp = tr.acts[l - 1] if l > 0 else tr.x
                add(gemm_key("wgrad", N, K), 2.0 * R * N * K,
                    lambda gl=gl, inp=inp, lin=lin: ops.linear_wgrad(gl, inp, dW=tr._gviews[id(lin.weight)], db=tr._gviews[id(lin.bias)],
                                                                     ws=tr.wgrad_ws))
    else:
        g_ = g_top
        for l in reversed(range(L)):
            lin = tr.linears[l]
            N, K = lin.out_features, lin.in_features
            inp = tr.acts[l - 1] if l > 0 else tr.x
            if wide_w and tr.wide_kinds[l] == 0:       # split-bf16 weight gradient on the plane copies the step left in place
                add(("linear_wgrad (wide layers)", "clica::wsplit::wgrad_split_k (+ slab_reduce_group_k)"), 2.0 * R * N * K,
                    lambda l=l: tr._wgrad_layer(l, None, None, tr.wgrad_ws, planes_ready=True))
            else:
                add(gemm_key("wgrad", N, K), 2.0 * R * N * K,
                    lambda g=g_, inp=inp, lin=lin: ops.linear_wgrad(g, inp, dW=tr._gviews[id(lin.weight)], db=tr._gviews[id(lin.bias)],
                                                                    ws=tr.wgrad_ws))
            if l > 0 and l in chain:
                prev = (l - 1) in chain
                out = None if prev else tr.dbuf[l & 1][:, :K]
                add(chain_key, 2.0 * R * N * K,
                    lambda lin=lin, l=l, prev=prev, out=out: ops.linear_split_dgrad(
                        tr.dzT[l], tr.wN[l], tr.xT[l], tr.slope, R, lin.out_features, lin.in_features,
                        dxT=tr.dzT[l - 1] if prev else None, dxN=tr.dzw_planes[l - 1] if tr.wide_kinds[l - 1] == 0 else None, dx=out,
                        **(dict(state=tr.s16, layer=l) if getattr(tr, "split_f16_wide", False) else {})))
                g_ = out if out is not None else g_
            elif l > 0:
                out = tr.dbuf[l & 1][:, :K]
                add(gemm_key("dgrad", N, K), 2.0 * R * N * K,
                    lambda g=g_, lin=lin, inp=inp, out=out: ops.linear_dgrad(g, lin.weight, inp, tr.slope, out=out))
                g_ = out
    # In-step timing of the fused encoder launches: the same launches, in the order and with the neighbours of the real step
    # (sample -> forward -> loss -> backward chain -> weight gradients -> Adam, eager on the launch stream), each bracketed by
    # HIP events.  This is the context the rocprofv3 kernel trace of the training loop averages over; a graph that replays ONE
    # symbol back to back sees a different L2 / Infinity-Cache state (its own 108 MB output is still resident) and the
    # forward / backward-chain pair alone thrashes differently again (measured 208 / 201 / 215 us for the three variants).
    instep = {}
    instep_ghz = {}
    instep_how = None
    if tr.fused_forward and tr.fused_backward and tr.grouped_wgrad and tr.head is None and not tr.dp and tr.graph is not None:
        # Preferred: the launches INSIDE the replayed step graph -- what the timed loop runs and what the rocprofv3 kernel trace
        # averages -- bracketed by device-side time stamps (clica_stamp: one-thread kernels writing the 100 MHz wall clock; event
        # records cannot be timed inside a captured graph).  The stamped graph is a second capture used for this leg only.
        names = ("mlp_fwd", "mlp_dgrad", "mlp_wgrad")
        WARM = 100
        snap = [t.clone() for t in (tr.param_arena, tr.exp_avg, tr.exp_avg_sq, tr.step_dev)]
        keep = tr.graph
        try:
            tr.stamps = {k: torch.zeros(1 + 2 * reps, dtype=torch.int64, device=tr.device) for k in names + ("null",)}
            tr.graph = None
            tr.capture()
            # shader clock over the same replays: the one-wave probe on a side stream (clica_clock_probe), 10 us between samples
            probe_stream = torch.cuda.Stream(device=tr.device)
            probe = ops.clock_probe((WARM + reps) * 60 + 2000, 10.0, probe_stream)       # ~0.6 ms per step + slack
            for _ in range(WARM + reps):
                tr.graph.replay()
            torch.cuda.synchronize()
            iv = {k: ops.stamp_intervals_us(tr.stamps[k]) for k in names + ("null",)}
            smp = probe.cpu().numpy()
            smp = smp[smp[:, 0] > 0]
            for k in names:
                g_ = [ops.clock_between(smp, b0, b1) for b0, b1 in ops.stamp_brackets(tr.stamps[k])]
                g_ = [x for x in g_ if x is not None and 0.2 < x < 3.0]
                if len(g_) >= max(3, reps // 4):
                    instep_ghz[k] = float(np.median(g_))
            if all(len(v) == reps for v in iv.values()):
                null_us = float(np.median(iv["null"]))       # begin stamp's run time + one launch boundary
                instep = {k: float(np.median(iv[k])) - null_us for k in names}
                instep_how = ("device time stamps (clica_stamp, 100 MHz wall clock) around the launch inside %d replays of the captured "
                              "training step, median, minus the %.2f us an empty stamp pair measures" % (reps, null_us))
        except Exception:
            instep = {}
        finally:
            tr.stamps = None
            tr.graph = keep
            for dst, src in zip((tr.param_arena, tr.exp_avg, tr.exp_avg_sq, tr.step_dev), snap):
                dst.copy_(src)
    if not instep and tr.fused_forward and tr.fused_backward and tr.grouped_wgrad and tr.head is None and not tr.dp:
        names = ("mlp_fwd", "mlp_dgrad", "mlp_wgrad")
        instep_how = "HIP events around the two launches inside eager training steps (see roofline_leg)"
        # one event set per repetition and NO host sync inside the loop: the host runs ahead of the GPU (a step is ~0.77 ms of GPU
        # work, ~0.3 ms of eager launch work), so every bracket opens while the GPU is still busy and measures kernel time
        WARM = 60      # ~45 ms of untimed steps first (sustained clock, see _graph_time)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(reps + WARM)]
        acc = {k: 0.0 for k in names}
        snap = [t.clone() for t in (tr.param_arena, tr.exp_avg, tr.exp_avg_sq, tr.step_dev)]
        for rep in range(reps + WARM):
            ev = evs[rep]
            tr._packed_current = False
            tr.sample()
            tr.pack()
            ev[0].record(); tr.forward(); ev[1].record()
            tr.loss_forward_backward()
            ev[2].record(); tr.backward_chain(g_top); ev[3].record()
            tr.weight_grads(g_top); ev[4].record()
            tr.optimizer_step()
        torch.cuda.synchronize()
        for rep in range(WARM, reps + WARM):
            ev = evs[rep]
            acc["mlp_fwd"] += ev[0].elapsed_time(ev[1]); acc["mlp_dgrad"] += ev[2].elapsed_time(ev[3]); acc["mlp_wgrad"] += ev[3].elapsed_time(ev[4])
        for dst, src in zip((tr.param_arena, tr.exp_avg, tr.exp_avg_sq, tr.step_dev), snap):
            dst.copy_(src)
        instep = {k: 1e3 * v / reps for k, v in acc.items()}       # us per launch (wgrad: its three launches together)
    rows = []
    for (op, sym), grp in groups.items():
        cnt = len(grp["fns"])
        if instep_how and instep_how.startswith("device time stamps") and (op in instep or op == fused_key[0]):
            # timed inside the replayed step: no isolated replays of these symbols (they would also enter the rocprofv3 average of
            # this command, which is meant to be the training loop's)
            sec = (instep[op] if op in instep else instep["mlp_fwd"] + instep["mlp_dgrad"]) * 1e-6
        else:
            sec = _graph_time(grp["fns"], reps)
        rows.append({"op": op, "kernel": sym, "launches_per_step": cnt, "avg_us": 1e6 * sec / cnt,
                     "gflop_per_launch": grp["flops"] / cnt / 1e9, "tflops": grp["flops"] / sec / 1e12, "us_per_step": 1e6 * sec})
    for r in rows:
        if r["op"] in instep:
            r["in_step_us"] = instep[r["op"]]
        if r["op"] in instep_ghz:
            r["shader_clock_ghz"] = round(instep_ghz[r["op"]], 3)      # the probe wave's cycles / wall time inside the op's brackets (median)
        elif r["op"] == fused_key[0] and "mlp_fwd" in instep_ghz and "mlp_dgrad" in instep_ghz:
            r["shader_clock_ghz"] = round(0.5 * (instep_ghz["mlp_fwd"] + instep_ghz["mlp_dgrad"]), 3)
    if instep:       # the dominant symbol's entry: average of its two in-step launches (forward stack + backward chain)
        for r in rows:
            if r["op"] == fused_key[0]:
                if not (instep_how or "").startswith("device time stamps"):
                    r["isolated_pair_avg_us"] = r["avg_us"]
                r["avg_us"] = 0.5 * (instep["mlp_fwd"] + instep["mlp_dgrad"])
                r["us_per_step"] = 2.0 * r["avg_us"]
                r["tflops"] = 2.0 * r["gflop_per_launch"] * 1e9 / (r["us_per_step"] * 1e-6) / 1e12
                r["timing"] = instep_how
    rows.sort(key=lambda r: -r["us_per_step"])
    peak = PEAK_FP32_MFMA_TFLOPS
    issued_factor = 1.0
    widths = [lin.out_features for lin in tr.linears]
    nparam = sum(lin.out_features * lin.in_features for lin in tr.linears)
    mask_b = (sum(m.numel() * 8 for m in tr.signmasks if m is not None) if tr.signmasks else 0)
    if fused_key in groups and split:
        # split-bf16 mode: the dominant symbol is the whole-stack kernel on the bf16 matrix cores (two launches per step).  Every
        # fp32 product is SIX bf16 MFMA products: `achieved` counts the bf16 flops the kernel ISSUES (6 x algorithmic) against the
        # dense bf16 peak; `fp32_equivalent` is the algorithmic 2MNK against the fp32 matrix peak the native kernel is bound by.
        top = [r for r in rows if r["op"] == fused_key[0]][0]
        f16 = bool(getattr(tr, "split_f16", False))
        pb = 4 if f16 else 6                       # bytes per element of a plane copy / of the piece weights
        peak, issued_factor = PEAK_BF16_MFMA_TFLOPS, (3.0 if f16 else 6.0)      # (the dense fp16 MFMA peak equals the bf16 one)
        pl = getattr(tr, "split_wgrad", False)
        per = lambda l, planes: (pb if (pl and planes[l] is not None) else 0) + (4 if (not pl or tr.acts_out[l] is not None) else 0)   # noqa: E731
        fwd_b = 4 * R * tr.linears[0].in_features * 2 + R * sum(widths[l] * per(l, tr.act_planes) for l in range(L)) + pb * nparam + mask_b
        perz = lambda l: (pb if (pl and tr.dz_planes[l] is not None) else 0) + (4 if (not pl or tr.dz_out[l] is not None) else 0)         # noqa: E731
        bwd_b = 4 * R * widths[-1] + R * sum(widths[l] * perz(l) for l in range(L - 1)) + pb * (nparam - widths[0] * tr.linears[0].in_features) + mask_b
        top["alg_bytes"] = (fwd_b + bwd_b) // 2
        note = (("f32 results from three fp16 products (hi.hi, hi.lo, lo.hi) of two-piece fp16 operand splits with per-tensor power-of-two "
                 "scales (v_mfma_f32_16x16x32_f16, fp32 accumulate); activation planes resident in LDS") if f16 else
                ("f32 results from six bf16 products of exact 3-way bf16 operand splits (v_mfma_f32_16x16x32_bf16, fp32 accumulate); "
                 "activation planes resident in LDS"))
    elif fused_key in groups:
        top = [r for r in rows if r["op"] == fused_key[0]][0]
        # minimum HBM bytes per launch (average of the two launches): every layer output written once (saved
        # activations / dZ), the weights once, the sign bits once, the 10-wide input
        fwd_b = 4 * R * (tr.linears[0].in_features + sum(widths)) + 4 * nparam + mask_b
        bwd_b = 4 * R * (widths[-1] + sum(widths[:-1])) + 4 * nparam + mask_b
        top["alg_bytes"] = (fwd_b + bwd_b) // 2
        note = "f32 (v_mfma_f32_16x16x4_f32), activation panel resident in LDS"
    elif chain_key in groups:
        # wide encoder with the split chain: the symbol with the largest share is the split GEMM of the 2000 x 2000 layers
        top = [r for r in rows if r["op"] == chain_key[0]][0]
        f16w = bool(getattr(tr, "split_f16_wide", False))
        peak, issued_factor = PEAK_BF16_MFMA_TFLOPS, (3.0 if f16w else 6.0)
        note = (("f32 results from three fp16 products of two-piece fp16 operand splits with per-tensor scales (v_mfma_f32_32x32x16_f16, fp32 accumulate); "
                 "operands as fp16 planes of the transposed tensors, fused bias / LeakyReLU / gate / re-split epilogue") if f16w else
                ("f32 results from six bf16 products of exact 3-way bf16 operand splits (v_mfma_f32_32x32x16_bf16, fp32 accumulate); "
                 "operands as bf16 planes of the transposed tensors, fused bias / LeakyReLU / gate / re-split epilogue"))
    else:
        top = [r for r in rows if r["op"] == "linear_fwd"][0]
        note = "f32 (v_mfma_f32_32x32x2_f32)"
    traffic, traffic_src = None, None
    try:        # HBM-side bytes per launch from the committed PMC passes (cannot be collected from inside this process)
        here = os.path.dirname(os.path.abspath(__file__))
        cand = sorted(f for f in os.listdir(os.path.join(here, "profiles")) if f.endswith("_traffic.json"))
        for fname in reversed(cand):             # newest round first; the file that knows this kernel symbol (headline / config-3 profiles)
            tj = json.load(open(os.path.join(here, "profiles", fname)))
            ent = tj["kernels"].get(top["kernel"].split(" (+")[0].split(" [")[0])
            if ent is None:         # per-layer symbols are abbreviated in the bench line ("...4 waves..."): match on the tile shape prefix
                pre = top["kernel"].split("...")[0]
                tail = top["kernel"].split("...")[-1].rstrip(">").strip(", ")
                hits = [v for k, v in tj["kernels"].items() if k.startswith(pre) and tail and k.rstrip(">").endswith(tail)] if "..." in top["kernel"] else []
                ent = hits[0] if len(hits) == 1 else None
            if ent:
                traffic = round(ent["fetch_x2_bytes"] + ent["write_bytes"])
                import hashlib
                digest = hashlib.sha256(open(os.path.join(here, "profiles", fname), "rb").read()).hexdigest()[:12]
                traffic_src = (f"NOT measured by this run: copied from the committed PMC profile profiles/{fname} (sha256 {digest}; "
                               "FETCH_SIZE x2 + WRITE_SIZE per launch, separate rocprofv3 --pmc passes of tools/profile_round.sh) -- "
                               "stale if the kernel changed after that profile was taken")
                break
    except Exception:
        pass
    # `achieved` / `frac`: ALGORITHMIC flops (2MNK per Linear application) / launch time against the peak of the matrix
    # pipe the kernel runs on; the flops the emulation ISSUES (3 or 6 piece products per fp32 product) are pipe
    # utilisation and sit beside it as `achieved_issued` / `frac_issued`
    roof = {"kernel": top["kernel"], "op": top["op"], "bound": "mfma", "achieved": round(top["tflops"], 2),
            "peak": peak, "unit": "TFLOP/s", "frac": round(top["tflops"] / peak, 4),
            "achieved_issued": round(issued_factor * top["tflops"], 2), "frac_issued": round(issued_factor * top["tflops"] / peak, 4),
            "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": top.get("alg_bytes"), "avg_launch_us": round(top["avg_us"], 2), "launches_per_step": top["launches_per_step"],
            "algorithmic_gflop_per_launch": round(top["gflop_per_launch"], 4), "dtype": note}
    if top.get("shader_clock_ghz"):
        ghz = top["shader_clock_ghz"]
        roof["shader_clock_ghz"] = ghz
        roof["frac_at_measured_clock"] = round(top["tflops"] / (peak * ghz / NOMINAL_GHZ), 4)
        roof["frac_issued_at_measured_clock"] = round(issued_factor * top["tflops"] / (peak * ghz / NOMINAL_GHZ), 4)
        roof["clock_note"] = ("shader clock over the kernel's in-step brackets (a one-wave probe on a side stream samples s_memtime / s_memrealtime every 10 us, clica_clock_probe); "
                              "`peak` is the nominal %.1f GHz figure, `frac_at_measured_clock` prices the same launch against the matrix rate "
                              "at the clock the chip actually held" % NOMINAL_GHZ)
    if issued_factor != 1.0:
        roof["flops_counted"] = ("`achieved` / `frac`: algorithmic 2MNK against the dense fp16 / bf16 MFMA peak; `*_issued`: x %d (the piece products one fp32 "
                                 "product costs in this arithmetic) = matrix-pipe utilisation; the emulation's own ceiling is peak / %d fp32-equivalent"
                                 % (int(issued_factor), int(issued_factor)))
        roof["fp32_equivalent"] = {"achieved": round(top["tflops"], 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(top["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4),
                                   "what": "algorithmic 2MNK per launch / launch time, against the fp32 matrix peak that bounds the native-fp32 kernel"}
    return roof, rows


def measure_traffic(args, kernel_symbol, timeout_s=150, child_args=None, grid_size=None):
    """HBM-side bytes per launch of `kernel_symbol`, measured BY THIS RUN: two short child runs of this script
    under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no other trace domains, as
    MI355X_MICROARCH.md prescribes), per-launch averages over the child's steps; FETCH_SIZE x 2 (the guide's gfx950 correction for wide
    coalesced reads) + WRITE_SIZE, both reported in KB.  Returns (bytes, description) or (None, reason)."""
    import csv, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    here = os.path.dirname(os.path.abspath(__file__))
    child = [sys.executable, os.path.join(here, "bench.py"), "--steps", "6", "--warmup", "2", "--windows", "1", "--no-cpu-baseline", "--no-roofline",
             "--no-native-leg", "--no-dropin", "--no-secondary", "--no-traffic", "--no-dry-leg", "--n", str(args.n), "--batch-size", str(args.batch_size),
             "--p", str(args.p), "--space-type", args.space_type] + (["--native-fp32"] if args.native_fp32 else [])
    if child_args is not None:      # another leg's command (bench.py --config c5 ...)
        child = [sys.executable, os.path.join(here, "bench.py")] + list(child_args)
    want = kernel_symbol.split(" (+")[0].split(" [")[0]
    vals = {}
    tmp = tempfile.mkdtemp(prefix="clica_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "-d", out, "-o", "p", "--output-format", "csv", "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            path = None
            for root, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(root, f)
            if r.returncode != 0 or path is None:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {(r.stderr or '')[-200:]}"
            acc = []
            for row in csv.DictReader(open(path)):
                if row.get("Counter_Name") == ctr and row["Kernel_Name"].split("(")[0].replace("void ", "") == want:
                    if grid_size is not None and str(row.get("Grid_Size", grid_size)) != str(grid_size):
                        continue      # (another launch shape of the same kernel)
                    acc.append(float(row["Counter_Value"]))
            if not acc:
                return None, f"no {ctr} rows for {want}"
            vals[ctr] = (sum(acc) / len(acc), len(acc))
    except Exception as e:      # noqa: BLE001  (a profiler hiccup must not cost the bench line)
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = 2.0 * vals["FETCH_SIZE"][0] * 1024 + vals["WRITE_SIZE"][0] * 1024
    return round(total), (f"measured by this run: child runs of bench.py (6 steps) under rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                          f"(separate passes); per-launch averages over {vals['FETCH_SIZE'][1]} / {vals['WRITE_SIZE'][1]} launches; 2 x FETCH_SIZE "
                          f"({2 * vals['FETCH_SIZE'][0] / 1024:.1f} MB) + WRITE_SIZE ({vals['WRITE_SIZE'][0] / 1024:.1f} MB)")


def loss_leg(tr, reps=20):
    """Event-time the tiled Lp-InfoNCE forward and backward (all their kernels) on the step's buffers."""
    import ctypes as C
    from cl_ica_amd import _lib
    lib, st = _lib.load(), _lib.stream_ptr()
    B, n, o = tr.B, tr.n, tr.loss_out
    y1, y2 = tr.y[:B], tr.y[B:]
    pooled = getattr(tr, "z_all", None) is not None          # data parallel, or the emulated / dry-run pool of an R-rank job
    z3 = tr.z_all if pooled else y1
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    train = getattr(tr, "loss_train", False)
    pool_lse = tr.lse_all if pooled else o[2 * B:3 * B]
    for _ in range(reps):
        ev[0].record()
        if train:      # the fused training pair of entry points the engine calls (coefficient step inside finalize, means inside the reduce)
            lib.clica_lp_loss_fwd_train(C.byref(tr.desc), y1.data_ptr(), n, y2.data_ptr(), n, z3.data_ptr(), n, o[:B].data_ptr(),
                                        o[B:2 * B].data_ptr(), o[2 * B:3 * B].data_ptr(), tr.dy[:B].data_ptr(), n, tr.dy[B:].data_ptr(), n,
                                        tr.loss_ws.data_ptr(), tr.loss_ws.numel(), st)
        else:
            lib.clica_lp_loss_fwd(C.byref(tr.desc), y1.data_ptr(), n, y2.data_ptr(), n, z3.data_ptr(), n, o[:B].data_ptr(),
                                  o[B:2 * B].data_ptr(), o[2 * B:3 * B].data_ptr(), o[3 * B:].data_ptr(), None, 0,
                                  tr.loss_ws.data_ptr(), tr.loss_ws.numel(), st)
        ev[1].record()
        if train:
            lib.clica_lp_loss_bwd_sym_train(C.byref(tr.desc), y1.data_ptr(), n, z3.data_ptr(), n, o[2 * B:3 * B].data_ptr(), pool_lse.data_ptr(),
                                            tr.dy[:B].data_ptr(), n, o[3 * B:].data_ptr(), None, tr.loss_ws.data_ptr(), tr.loss_ws.numel(), st)
        else:
            lib.clica_lp_loss_bwd_sym(C.byref(tr.desc), y1.data_ptr(), n, y2.data_ptr(), n, z3.data_ptr(), n, o[2 * B:3 * B].data_ptr(),
                                      pool_lse.data_ptr(), None, None, None,
                                      tr.dy[:B].data_ptr(), n, tr.dy[B:].data_ptr(), n, tr.loss_ws.data_ptr(), tr.loss_ws.numel(), st)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]) * 1e-3; tb += ev[1].elapsed_time(ev[2]) * 1e-3
    pairs = float(B) * z3.shape[0] + B
    path = C.c_int32(0)
    if train:
        lib.clica_lp_loss_train_path(C.byref(tr.desc), C.byref(path))
    cp = {1: 2, 2: 2, 3: 4}.get(int(tr.p), 6)
    fl_f = pairs * (cp * n + 6)      # P (c_p n + 6); backward contract figure = 3x forward
    # (the symmetric backward does ONE pair sweep; the contract's 3x is kept as the algorithmic figure)
    return {"fwd_us": 1e6 * tf / reps, "bwd_us": 1e6 * tb / reps, "pairs": pairs,
            "fwd_gpairs_per_s": pairs / (tf / reps) / 1e9, "fwd_tflops_valu": fl_f / (tf / reps) / 1e12,
            "bwd_tflops_valu": 3 * fl_f / (tb / reps) / 1e12, "valu_peak_tflops": PEAK_FP32_VALU_TFLOPS,
            "fwd_valu_frac": fl_f / (tf / reps) / 1e12 / PEAK_FP32_VALU_TFLOPS, "bwd_valu_frac": 3 * fl_f / (tb / reps) / 1e12 / PEAK_FP32_VALU_TFLOPS,
            "negatives_pool": int(z3.shape[0]),
            "algorithmic_bytes_fwd": 4 * n * (2 * B + z3.shape[0]) + 12 * B,
            "sweeps": ("bf16 matrix cores (csrc/lp_mfma.hip: logit = augmented inner product of exact 3-piece splits, gradient = second product "
                       "against the pool; the *_valu figures keep SURVEY 8(d)'s contract count as the algorithmic work)" if path.value == 1
                       else "vector ALU on coordinate differences (csrc/lp_kernels.h)")}


def dropin_leg(args, device, steps=40, warmup=8):
    """The INTEGRATION.md import swap, measured: the reference's own `train_step` structure (main_mlp.py:258-285 -- two h(z)
    passes, roll inside the graph, loss(...), backward(), optimizer.step(), `.item()` host syncs) and its sampling call
    (:196-200, :328) with `import cl_ica_amd.{losses,encoders,latent_spaces,spaces,invertible_network_utils}` in place of the
    reference's modules; torch autograd drives the HIP kernels through the drop-in modules.  Reported next to the fused engine:
    `torch_adam` keeps the reference's `torch.optim.Adam` line, `flat_adam` swaps that one line for `cl_ica_amd.optim.Adam`,
    `captured` additionally wraps the closure once with `cl_ica_amd.capture_train_step` (one graph launch per step);
    `torch_adam_captured` does that with `torch.optim.Adam(..., capturable=True)` (the encoder then stays on bf16x3: the f16x2
    arithmetic needs the flat Adam's launch for its guard)."""
    import contextlib, io, types
    import cl_ica_amd
    from cl_ica_amd import encoders, invertible_network_utils as inu, lazy, losses, optim, train_mlp
    lazy_on = lazy.enabled()
    n, B = args.n, args.batch_size
    a = types.SimpleNamespace(n=n, box_min=0.0, box_max=1.0, sphere_r=1.0, m_param=1.0, m_p=0, c_param=0.05, c_p=2,
                              space_type=args.space_type)
    latent_space = train_mlp.build_latent_space(a, train_mlp.sampler_spec(a, 0))
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        g = inu.construct_invertible_mlp(n=n, n_layers=3, act_fct="leaky_relu", cond_thresh_ratio=0.0,
                                         n_iter_cond_thresh=25000 if n <= 10 else 2000).to(device)
    loss = losses.LpSimCLRLoss(p=args.p, tau=1.0, simclr_compatibility_mode=True)
    res = {}
    for name in ("torch_adam", "flat_adam", "torch_adam_captured", "captured"):
        torch.manual_seed(0)
        f = encoders.get_mlp(n_in=n, n_out=n, layers=[n * 10, n * 50, n * 50, n * 50, n * 50, n * 10]).to(device)
        optimizer = (torch.optim.Adam(f.parameters(), lr=1e-4) if name == "torch_adam" else
                     torch.optim.Adam(f.parameters(), lr=1e-4, capturable=True) if name == "torch_adam_captured" else optim.Adam(f.parameters(), lr=1e-4))
        h = lambda z: f(g(z))   # noqa: E731

        def train_step(data, loss, optimizer):
            z1, z2_con_z1 = data
            z3 = torch.roll(z1, 1, 0)
            optimizer.zero_grad()
            z1_rec = h(z1)
            z2_con_z1_rec = h(z2_con_z1)
            z3_rec = torch.roll(z1_rec, 1, 0)
            total_loss_value, _, losses_value = loss(z1, z2_con_z1, z3, z1_rec, z2_con_z1_rec, z3_rec)
            total_loss_value.backward()
            optimizer.step()
            return total_loss_value.item(), [v.item() for v in losses_value]

        if name.endswith("captured"):       # the same closure, recorded once into a HIP graph (cl_ica_amd/graphed.py); the call site below is unchanged
            try:
                z = latent_space.sample_marginal(B)
                train_step = cl_ica_amd.capture_train_step(train_step, (z, latent_space.sample_conditional(z, B)), loss, optimizer)
            except Exception as e:   # noqa: BLE001  (reported, not hidden: the two eager legs above are the fallback a user has)
                res[name] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
                continue

        def one():
            z = latent_space.sample_marginal(B)
            return train_step((z, latent_space.sample_conditional(z, B)), loss, optimizer)
        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            lv, _ = one()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ar = encoders.arith_state(f)
        res[name] = {"value": steps / el, "unit": "steps/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "final_loss": lv,
                     "encoder_arith": ar["arith"], **({"f16_steps_withheld": ar["skipped"], "f16_flags": ar["flags"]} if "skipped" in ar else {})}
    st = f._structure() if hasattr(f, "_structure") else (None,) * 5 + (False, False)
    fused = encoders._use_fused(st[5], 2 * B if lazy_on else B)
    res["encoder_path"] = (("whole-encoder kernels in the split-bf16 arithmetic (clica_mlp_fwd_split / clica_mlp_dgrad_split / "
                            "clica_mlp_wgrad_split, or their f16x2 forms *16 under cl_ica_amd.optim.Adam -- `encoder_arith` of each leg; weight "
                            "gradients added into the flat optimizer's gradient arena in place)"
                            if encoders._dropin_split(st[6]) else
                            "whole-encoder kernels (clica_mlp_fwd / clica_mlp_dgrad / clica_mlp_wgrad)") if fused else
                           "per-layer GEMM kernels (clica_linear_*): a %d-row encoder call is %d workgroups of 48 rows, below the 128 the "
                           "whole-encoder kernels need to beat them" % (B, (B + 47) // 48))
    res["what"] = ("reference train_step structure (main_mlp.py:258-285: two encoder calls of B rows, roll in the graph, three .item() "
                   "calls per step, eager launches, torch autograd) on the drop-in modules; deferred stacking of the two encoder calls "
                   f"{'on' if lazy_on else 'off'} (CLICA_DROPIN_LAZY), roll detection -> one-sweep symmetric loss backward, loss scalars by one "
                   "async copy behind the loss forward, weights re-packed at step end")
    return res


def capture_or_eager(tr, args, rank, world, device):
    """HIP-graph capture of the step (all ranks agree on the outcome); returns whether replays are in use."""
    use_graph = not args.no_graph
    if use_graph:
        try:
            inj = os.environ.get("CLICA_BENCH_INJECT_CAPTURE_FAILURE")      # test hook: "all" or a rank number (tests/test_gpu_bench.py)
            if inj is not None and world > 1 and inj in ("all", str(rank)):
                raise RuntimeError("injected capture failure (CLICA_BENCH_INJECT_CAPTURE_FAILURE)")
            tr.capture()
        except Exception as e:   # e.g. a RCCL build that cannot capture collectives: run eagerly, say so
            if world == 1:
                raise
            print(f"[bench] graph capture with collectives failed on rank {rank} ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            use_graph = False
            tr.graph = None
        if world > 1:
            ok = torch.tensor([1 if use_graph else 0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                use_graph = False
                tr.graph = None
    return use_graph


def timed_windows(tr, steps, warmup, windows, world, device):
    """`warmup` untimed steps (+ >= 60 ms of continuous work: the chip settles at its sustained clock only after tens of
    milliseconds, see _graph_time), then `windows` windows of EXACTLY `steps` steps, every window bracketed by barrier +
    synchronize on both sides and reduced with MAX over the ranks.  Returns (window seconds, extra warm-up steps)."""
    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)   # (--launch-check runs on CPU ranks too)
    for _ in range(warmup):
        tr.step()
    sync()
    extra_warm = 0
    if world == 1:
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.06:
            for _ in range(10):
                tr.step()
            extra_warm += 10
            sync()
    else:               # a FIXED count under data parallelism: every rank must run the same number of collective-bearing steps
        extra_warm = max(0, 80 - warmup)
        for _ in range(extra_warm):
            tr.step()
    window_s = []
    for _ in range(max(1, windows)):
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step()
        sync()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        window_s.append(el)
    return window_s, extra_warm


def comm_leg(tr, rank, world, device, reps=20):
    """N > 1 only: what every rank saw of the job -- RCCL world size, and the wall time of each collective of the step in
    isolation (HIP events on the launch stream, barrier in front of every repetition): the all-gather of the embeddings, the
    all-gather of the row statistics, the all-reduce of the flat gradient arena.  Gathered to rank 0 so that the first real
    multi-GPU run explains its own scaling curve."""
    B, n = tr.B, tr.n
    y1 = tr.y[:B].contiguous()
    lse = tr.loss_out[2 * B:3 * B]
    legs = {"all_gather_embeddings": lambda: dist.all_gather_into_tensor(tr.z_all, y1, group=tr.pg),
            "all_gather_row_lse": lambda: dist.all_gather_into_tensor(tr.lse_all, lse, group=tr.pg),
            "all_reduce_grad_arena": lambda: dist.all_reduce(tr.grad_arena, group=tr.pg)}
    mine = {"rank": rank, "world_size_seen": dist.get_world_size(), "backend": dist.get_backend(), "device": torch.cuda.get_device_name(device)}
    grad_snapshot = tr.grad_arena.clone()
    for name, fn in legs.items():
        for _ in range(3):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
        for r in range(reps):
            dist.barrier()
            ev[2 * r].record(); fn(); ev[2 * r + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[2 * r].elapsed_time(ev[2 * r + 1]) * 1e3 for r in range(reps))
        mine[name + "_us"] = round(ts[len(ts) // 2], 1)
    tr.grad_arena.copy_(grad_snapshot)
    mine["bytes"] = {"all_gather_embeddings": 4 * B * n, "all_gather_row_lse": 4 * B, "all_reduce_grad_arena": 4 * tr.grad_arena.numel()}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    return allr


def secondary_leg(args, device, steps=20, windows=3):
    """BASELINE config 3 on ONE rank (main_mlp.py --n 40 --space-type sphere --p 1 --batch-size 6144: 13.6 M parameters, 1.005
    TFLOP of encoder GEMMs per step, 2000-wide layers -> per-layer kernels, the 2000 x 2000 layers in split-bf16): step rate with the local negatives
    pool (B3 = 6144) and with the pool of the 8-GPU job emulated on this GPU (B3 = 49 152: the local embeddings replicated eight
    times, device copies in place of the two all-gathers -- this rank's compute of the data-parallel job, no communication)."""
    import copy
    a = copy.copy(args)
    a.n, a.space_type, a.p = 40, "sphere", 1
    res = {"workload": "main_mlp.py --n 40 --space-type sphere --p 1 --batch-size 6144 (BASELINE configs[2], one rank's work)",
           "encoder_gflop_per_step": round(3 * 2 * 13632000 * 2 * a.batch_size / 1e9, 1),
           "dtype": None}
    for name, ranks in (("pool_6144", 1), ("pool_49152_emulated_8_ranks", 8)):
        tr = build_trainer(a, device, 1, emulate_pool_ranks=ranks)
        sp = "f16x2 split (3 fp16 MFMA products, per-tensor scales)" if getattr(tr, "split_f16_wide", False) else "bf16x3 split"
        res["dtype"] = (f"f32: the 2000 x 2000 layers via {sp} (forward, data and weight gradients: gemm_split_k / wgrad_split_k); narrow "
                        f"layers native fp32 MFMA forward / data gradient, {sp} weight gradients")
        capture_or_eager(tr, a, 0, 1, device)
        w, _ = timed_windows(tr, steps, 5, windows, 1, device)
        el = float(np.median(w))
        ent = {"value": steps / el, "unit": "steps/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "windows": len(w),
               "negatives_pool": a.batch_size * ranks, "final_loss": float(tr.loss_out[3 * tr.B].item()),
               "whole_step_encoder_tflops": round(3 * 2 * 13632000 * 2 * a.batch_size / (el / steps) / 1e12, 1)}
        if ranks == 1 and not args.no_roofline:
            roof, rows = roofline_leg(tr, reps=5)
            ent["roofline"] = {k: roof[k] for k in ("kernel", "op", "achieved", "peak", "unit", "frac", "achieved_issued", "frac_issued", "avg_launch_us",
                                                    "launches_per_step", "algorithmic_gflop_per_launch", "dtype")}
            ent["kernels"] = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]
            ll = loss_leg(tr, reps=5)
            ent["loss_kernel"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ll.items()}
        res[name] = ent
        del tr
        torch.cuda.empty_cache()
    return res


def conv_config_leg(which, device, steps=20, warmup=6, windows=3):
    """BASELINE configs[3] / configs[4] on one GPU: the conv encoders run on PyTorch-ROCm / MIOpen as north_star
    prescribes, head / loss / optimizer on the HIP library, the training step is the reference's own.
      c4  main_3dident.py:467-503 train_step: ResNet-18 (cl_ica_amd/resnet.py, torchvision's layout) on (1024, 3, 64, 64) x 2 views,
          NCHW like the reference, BatchNorm on batch statistics, LeakyReLU -> Linear(30 -> 3) -> position-only box head, LpSimCLRLoss(p = 2), flat Adam
      c5  kitti_masks/solver.py:61-74: BetaVAE_H (five k = 4 convs + Linear(256 -> 5)) on (2048, 1, 64, 64) Bernoulli(0.1) masks = 1024
          pairs, z_dim 5, p = 1, flat Adam; the whole global batch of the 4-rank job on ONE GPU (per rank it is 512 images / 256 pairs)
    Synthetic inputs of the reference's shapes (no dataset in the image), no host sync inside the timed windows."""
    import types
    from cl_ica_amd.optim import Adam
    torch.manual_seed(0)
    if which == "c4":
        from cl_ica_amd import threedident as T
        a = types.SimpleNamespace(position_only=True, rotation_and_color_only=False, rotation_only=False, color_only=False,
                                  non_periodic_rotation_and_color=False, box_constraint="fix", sphere_constraint=None,
                                  unsupervised_loss="l2", identity_solution=False, encoder="rn18")
        # NCHW, the reference's layout (main_3dident.py never sets a memory format): measured 31.7 steps/s against 21.1 in channels_last,
        # where MIOpen's NHWC BatchNorm kernels take 53 % of the step (tools/conv_layout_probe.py, profiles/r4_c4_*)
        f = T.setup_f(a, 3, 0).to(device)
        f.train()
        loss = T.make_unsupervised_loss(a, 3)
        opt = Adam(f.parameters(), lr=1e-4)
        x1 = torch.randn(1024, 3, 64, 64, device=device)
        x2 = x1 + 0.1 * torch.randn_like(x1)

        def step():
            return T.train_step(((None, None), (x1, x2)), loss, opt, f, sync=False)[0]
        n_params = sum(p.numel() for p in f.parameters())
        work = ("main_3dident.py train_step (:467-503): ResNet-18 backbone (MIOpen, NCHW) on (1024, 3, 64, 64) x 2 views -> HIP "
                "LeakyReLU / Linear(30, 3) / Softclip head -> HIP LpSimCLRLoss(p = 2) -> backward -> flat HIP Adam")
    else:
        from cl_ica_amd.kitti_masks.solver import Solver
        a = types.SimpleNamespace(cuda=True, ckpt_dir="/tmp", output_dir="/tmp", dataset="kitti", max_iter=1, z_dim=5, num_channel=1,
                                  lr=1e-4, beta1=0.9, beta2=0.999, ckpt_name="last", log_step=1000, save_step=10 ** 9, box_norm=True, p=1)
        S = Solver(a, None)
        S.net_mode(train=True)
        x = (torch.rand(2048, 1, 64, 64, device=device) < 0.1).float()

        launch_mode = "eager (torch autograd drives the HIP library)"
        step_eager = lambda: S.train_iteration(x)      # noqa: E731
        step = step_eager
        if os.environ.get("CLICA_C5_GRAPH", "1") != "0" and os.environ.get("CLICA_CONV", "hip") != "miopen":
            try:      # the whole iteration as one HIP graph (Solver.capture); any failure falls back to the eager loop and is reported
                replay, loss_t = S.capture(x)

                def step():
                    replay()
                    return loss_t
                launch_mode = "HIP graph replay of the whole iteration (Solver.capture: forward, loss, backward, flat Adam)"
            except Exception as e:      # noqa: BLE001
                step = step_eager
                launch_mode = f"eager (graph capture failed: {type(e).__name__}: {str(e)[:160]})"
        n_params = sum(p.numel() for p in S.net.parameters())
        conv_path = ("MIOpen via nn.Conv2d (CLICA_CONV=miopen)" if os.environ.get("CLICA_CONV", "hip") == "miopen"
                     else "HIP implicit-GEMM stages, clica_conv16_* in the f16x2 split arithmetic (CLICA_CONV_ARITH=f32: the fp32-MFMA kernels; CLICA_CONV=miopen: nn.Conv2d)"
                     if os.environ.get("CLICA_CONV_ARITH", "f16x2") == "f16x2" else "HIP implicit-GEMM stages, clica_conv_* on fp32 MFMA (CLICA_CONV_ARITH=f32)")
        work = ("kitti_masks Solver.train body (solver.py:61-74): BetaVAE_H conv encoder (%s) on (2048, 1, 64, 64) binary masks = 1024 "
                "pairs -> HIP Linear(256, 5) / Softclip -> strided views -> HIP LpSimCLRLoss(p = 1) -> backward -> flat HIP Adam" % conv_path)
    for _ in range(warmup):
        last = step()
    torch.cuda.synchronize(device)
    ws = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            last = step()
        e1.record()
        torch.cuda.synchronize(device)
        ws.append(e0.elapsed_time(e1) * 1e-3)
    el = float(np.median(ws))
    extra = {}
    if which == "c4":
        # config 4's step is MIOpen's (ResNet-18 backbone on PyTorch-ROCm, as north_star allows): its `roofline` is a
        # whole-step estimate -- ResNet-18 at 64 x 64 is (64 / 224)^2 x 1.82 GMAC = 0.149 GMAC per image forward, x 3 for forward + both
        # backward products, 2 x 1024 images per step -- against the fp32 matrix peak the fp32 NCHW convolutions could reach at best
        gflop_step = 2.0 * 0.149 * 3.0 * 2048
        extra["roofline"] = {"kernel": "MIOpen convolution kernels (ResNet-18 backbone; none of this repository's kernels: head, loss and Adam are < 2 % of the step)",
                             "bound": "mfma", "achieved": round(gflop_step * steps / el / 1e3, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(gflop_step * steps / el / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                             "what": "WHOLE-STEP ESTIMATE, not a kernel measurement: 1 831 algorithmic GFLOP per step (0.149 GMAC / image forward x 3 x 2048 "
                                     "images) / step time; the number says how MIOpen's fp32 NCHW path uses the chip, nothing about this repository's kernels",
                             "traffic": None}
    return {**extra, "workload": work, "value": steps / el, "unit": "steps/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "windows": windows,
            "images_per_s": round((2048 if which == "c5" else 2048) * steps / el, 1), "parameters": int(n_params),
            "dtype": ("f32 via f16x2 split in the 16C-deep conv stages (3 fp16 MFMA products of two-piece operands, fp32 accumulate), f32 elsewhere"
                      if which == "c5" and os.environ.get("CLICA_CONV", "hip") != "miopen" and os.environ.get("CLICA_CONV_ARITH", "f16x2") == "f16x2" else "f32"),
            "final_loss": float(last.item()), "launch": (launch_mode if which == "c5" else "eager (torch autograd drives the HIP library and MIOpen)"),
            "kernel_shares": "profiles/%s_%s_summary.md (rocprofv3 --kernel-trace --stats of `bench.py --config %s`)" % ("r5" if which == "c5" else "r4", which, which)}


def c5_conv_roofline(device, reps=20, traffic=True):
    """`roofline` object of BASELINE configs[4]'s dominant kernel.  Round 5: the stack runs in the f16x2 split
    arithmetic (csrc/conv16.hip) and the longest launch is the FORWARD of the widest stage (32 -> 32 channels, 2048 x 16 x 16 output pixels,
    K = 512: `conv16::fwd_tile16_k`).  Timed by HIP events around isolated launches on the buffers one real forward / backward of the
    2048-mask batch has left behind (the step itself is eager torch autograd, its launches cannot be bracketed from here).  By the roofline
    model the kernel is HBM-bound: 17.2 GFLOP over 370 MB (S1 read once + S2 written) = 46 flop/B against a balance of 104 flop/B for three
    fp16 products per fp32 product -- so `achieved` / `peak` are algorithmic bytes per second against 8 TB/s; the matrix-side figures
    (issued fp16 flops against the 2.5 PFLOP/s pipe the kernel runs on, fp32-equivalent flops against the fp32 matrix peak that bounded
    the round-4 kernel) ride along.  `traffic` from child runs of `bench.py --config c5` under rocprofv3 --pmc (separate passes)."""
    import ctypes as C
    from cl_ica_amd import conv, _lib
    from cl_ica_amd.kitti_masks.model import BetaVAE_H
    lib = _lib.load()
    torch.manual_seed(0)
    net = BetaVAE_H(z_dim=5, nc=1, box_norm=True).to(device)
    x = (torch.rand(2048, 1, 64, 64, device=device) < 0.1).float()
    conv._POOL.clear()
    net(x).sum().backward()
    torch.cuda.synchronize(device)
    buf = conv._POOL[(2048, 1, device.index if device.index is not None else torch.cuda.current_device())][0]
    images, l = 2048, 1
    cout, ho = conv.STAGES[l]; cin, hs = conv.STAGES[l - 1][0], ho + 1
    st = _lib.stream_ptr()
    f16 = conv.get_arith() == "f16x2"
    bias = net.encoder[2].bias.detach()

    def launch():
        if f16:
            _lib.check(lib.clica_conv16_k4s2_fwd(buf.S[l].data_ptr(), buf.w16[l - 1].data_ptr(), buf.wscale.data_ptr() + 4 * (l - 1), bias.data_ptr(),
                                                 images, cin, cout, hs, hs, 1, 1, buf.S[l + 1].data_ptr(), buf.gate[l].data_ptr(),
                                                 conv._slots(buf, l - 1), conv._slots(buf, l), st), "clica_conv16_k4s2_fwd")
        else:
            _lib.check(lib.clica_conv_k4s2_fwd(buf.S[l].data_ptr(), buf.wpack[l].data_ptr(), bias.data_ptr(), images, cin, cout, hs, hs,
                                               1, 1, buf.S[l + 1].data_ptr(), buf.gate[l].data_ptr(), st), "clica_conv_k4s2_fwd")
    for _ in range(5):
        launch()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        launch()
    ev[1].record()
    torch.cuda.synchronize(device)
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    gflop = 2.0 * images * ho * ho * 16 * cin * cout / 1e9
    alg_bytes = int(4 * images * (hs * hs * 4 * cin + (ho // 2 + 1) ** 2 * 4 * cout) + images * ho * ho * cout // 8)
    tbs = alg_bytes / us / 1e6
    roof = {"kernel": "clica::conv16::fwd_tile16_k" if f16 else "clica::gemm::conv_gemm_k<128, 32, 4, 1, 2, true, true, 0>",
            "op": "forward of the 32 -> 32 stage (kitti_masks/model.py:41-56, second Conv2d + ReLU): implicit GEMM 524 288 x 32 x 512, bias + ReLU + "
                  "scatter into the next stage's input + gate bits + maximum in the epilogue",
            "bound": "hbm", "achieved": round(tbs * 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(tbs * 1e3 / PEAK_HBM_GBS, 4),
            "avg_launch_us": round(us, 2), "timing": f"HIP events around {reps} isolated launches on the 2048-mask batch's buffers",
            "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_gflop_per_launch": round(gflop, 3),
            "matrix_side": {"issued_tflops": round((3.0 if f16 else 1.0) * gflop / us * 1e3, 1), "pipe_peak_tflops": PEAK_F16_MFMA_TFLOPS if f16 else PEAK_FP32_MFMA_TFLOPS,
                            "frac_of_the_pipe_it_runs_on": round((3.0 if f16 else 1.0) * gflop / us * 1e3 / (PEAK_F16_MFMA_TFLOPS if f16 else PEAK_FP32_MFMA_TFLOPS), 4),
                            "fp32_equivalent_tflops": round(gflop / us * 1e3, 1), "frac_of_fp32_matrix_peak": round(gflop / us * 1e3 / PEAK_FP32_MFMA_TFLOPS, 4)},
            "dtype": ("f32 results from three fp16 products of two-piece operand splits, per-tensor power-of-two scales measured in the same step "
                      "(v_mfma_f32_32x32x16_f16, fp32 accumulate)") if f16 else "f32 (v_mfma_f32_32x32x2_f32, exact fp32 products)",
            "what_bounds_it": ("HBM: a workgroup fetches the 9 x 17 input pixels of its 8 x 16 output pixels once (one contiguous 78 KB block, split to f16 pieces "
                               "on the way into LDS, weights in registers), so the launch moves its algorithmic bytes ~once; the streaming version of round 5 "
                               "(every output pixel fetching its own 2 x 2 window: 1.07 GB through the L1) took 148 us") if f16 else "fp32 matrix rate at N = 32",
            "traffic": None, "traffic_source": None}
    class _A:      # what measure_traffic reads of the headline's arguments (unused by the c5 child command)
        n, batch_size, p, space_type, native_fp32 = 10, 6144, 2, "box", False
    if traffic:
        tb, src = measure_traffic(_A, roof["kernel"], child_args=["--config", "c5", "--steps", "3"])
        roof["traffic"], roof["traffic_source"] = tb, src
    if f16:
        # the kernel with the LARGEST share of config 5's step is the streaming kernel `stream16_k<4>` (21 % of GPU time,
        # profiles/r5_c5_summary.md), and its slowest instance is the second stage's DATA gradient (dO of the 32-channel 16 x 16 stage ->
        # dO of the 32-channel 32 x 32 stage: 268 MB of stores).  That launch is the config's `roofline`; the tile kernel above rides along.
        l = 1
        cout, ho = conv.STAGES[l]; cin, hs = conv.STAGES[l - 1][0], ho + 1
        dgrid = conv.STAGES[l - 1][1]
        sd = 3 + (3 - l)

        def launch_dg():
            _lib.check(lib.clica_conv16_k4s2_dgrad(buf.dO[l].data_ptr(), buf.w16[3 + l - 1].data_ptr(), buf.wscale.data_ptr() + 4 * (3 + l - 1), images, cin, cout,
                                                   hs, hs, buf.dO[l - 1].data_ptr(), dgrid, dgrid, buf.gate[l - 1].data_ptr(),
                                                   conv._slots(buf, sd), None, st), "clica_conv16_k4s2_dgrad")
        for _ in range(5):
            launch_dg()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            launch_dg()
        ev[1].record()
        torch.cuda.synchronize(device)
        us_d = ev[0].elapsed_time(ev[1]) * 1e3 / reps
        # algorithmic bytes: dO of this stage read once (padded grid hs x hs, cout channels), dO of the stage below written once
        # (dgrid x dgrid pixels, cin channels), one gate bit per written element
        alg_d = int(4 * images * (hs * hs * cout + dgrid * dgrid * cin) + images * dgrid * dgrid * cin // 8)
        gflop_d = 2.0 * images * hs * hs * (4 * cout) * (4 * cin) / 1e9
        roof_d = {"kernel": "clica::conv16::stream16_k<4>",
                  "op": "data gradient of the 32 -> 32 stage (autograd of kitti_masks/model.py:41-56, second Conv2d): dS = dO Wd as a streaming GEMM "
                        "591 872 x 128 x 128 with the ReLU gate and the scatter into the lower stage's gradient grid in the epilogue",
                  "bound": "hbm", "achieved": round(alg_d / us_d / 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(alg_d / us_d / 1e3 / PEAK_HBM_GBS, 4),
                  "avg_launch_us": round(us_d, 2), "timing": f"HIP events around {reps} isolated launches on the 2048-mask batch's buffers",
                  "algorithmic_bytes_per_launch": alg_d, "algorithmic_gflop_per_launch": round(gflop_d, 3),
                  "share": "largest share of the step's GPU time among config 5's kernels (stream16_k<4>: 21 %, profiles/r5_c5_summary.md); this is its slowest instance",
                  "what_bounds_it": "operand loads with two waves per SIMD (234 registers per wave): the wave waits 0.44 of its cycles, the matrix pipes are 0.18 busy "
                                    "(tools/c5_pmc_dispatch.sh); the stores alone are 268 MB",
                  "traffic": None, "traffic_source": None, "forward_tile_kernel": roof}
        if traffic:
            tb, src = measure_traffic(_A, roof_d["kernel"], child_args=["--config", "c5", "--steps", "3"])
            roof_d["traffic"], roof_d["traffic_source"] = tb, (src or "") + " [per-launch average over ALL launches of the symbol in the step: three stages' forward and data gradients]"
        roof = roof_d
    del net
    conv._POOL.clear()
    torch.cuda.empty_cache()
    return roof


def dry_ranks_leg(args, device, R=8, steps=50, windows=3):
    """Rank 0 of an R-rank data-parallel job, planned, captured (with its RCCL collectives, on a one-rank group) and run on THIS GPU:
    pool of R x B rows, loss workspaces / stream splits for that pool, two-half weight-gradient launch, gradient buckets, 1 / R
    gradient scale (ContrastiveTrainer(dry_ranks=R)).  What it shows: this rank's compute at R GPUs, i.e. the
    ceiling of per-rank speed before any wire time.  What it cannot show: the wire.  NOT a multi-GPU measurement."""
    made_group = False
    try:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            made_group = True
        trd = build_trainer(args, device, 1, dry_ranks=R)
        use_graph = capture_or_eager(trd, args, 0, 1, device)
        w, _ = timed_windows(trd, steps, args.warmup, windows, 1, device)
        el = float(np.median(w))
        loss = loss_leg(trd, reps=20)
        plan = trd.plan_summary()
        coll = plan.get("collectives_per_step", [])
        return {"what": f"DRY RUN: rank 0 of a {R}-rank job on one GPU (B={args.batch_size}/rank, n={args.n}; the other ranks' rows are copies) -- this "
                        "rank's compute with the R-rank negatives pool and every collective issued on a one-rank RCCL group inside the step graph; "
                        "not a multi-GPU measurement",
                "value": steps / el, "unit": "steps/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "windows": windows,
                "launch": "hipGraph replay with the RCCL collectives captured" if use_graph else "eager",
                "loss_fwd_us": round(loss["fwd_us"], 1), "loss_bwd_us": round(loss["bwd_us"], 1), "loss_sweeps": loss.get("sweeps"),
                "negatives_pool": plan.get("pool_rows"), "collective_bytes_per_step": sum(int(c.get("bytes_per_rank", c.get("bytes", 0))) for c in coll),
                "plan": plan, "final_loss": float(trd.loss_out[3 * trd.B].item())}
    except Exception as e:      # (a RCCL build that refuses a one-rank group / capture must not take the headline line down)
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        if made_group:
            try:
                dist.destroy_process_group()
            except Exception:
                pass


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves (one process per GPU,
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 and a free port) with the same arguments, pass
    their stdout / stderr through (rank 0's JSON line stays the last line of stdout) and return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # `--` ends torchrun's own options: bench.py's `--n` would otherwise be claimed as an abbreviation of torchrun's `--nnodes`
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--", os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks on 127.0.0.1:{port}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def launch_check(args, rank, world, device):
    """`--launch-check`: the multi-rank control flow of this file WITHOUT the engine -- rendezvous, `timed_windows` (barrier +
    synchronize on both sides of every window, MAX over the ranks) around a step that is one small all-reduce, process group torn
    down and C stdio flushed before rank 0 prints the contract-shaped line.  Runs on CPU ranks over gloo as well (tests/
    test_host_logic.py::test_bench_self_launch_two_ranks); `value` is null: nothing is measured."""
    class _Stub:
        def __init__(self):
            self.t = torch.ones(8, device=device)

        def step(self):
            if world > 1:
                dist.all_reduce(self.t)
                self.t /= world
    stub = _Stub()
    window_s, extra_warm = timed_windows(stub, args.steps, args.warmup, args.windows, world, device)
    seen = torch.tensor([1.0], device=device)
    if world > 1:
        dist.all_reduce(seen)
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        print(json.dumps({"metric": "launch check (no engine, nothing measured)", "value": None, "unit": "steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "none", "config": {"workload": "launch check", "parallelism": f"dp{world}"},
                          "launch_check": {"ranks_seen": int(seen.item()), "windows": len(window_s), "device": device.type,
                                           "value_all_reduced": float(stub.t[0].item())}}), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    from cl_ica_amd.distributed import init_from_env
    rank, world, device = init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or unset WORLD_SIZE: bench.py then starts the ranks itself)")
    if args.launch_check:
        return launch_check(args, rank, world, device)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.dry_ranks and args.dry_ranks > 1:
        if world != 1:
            raise SystemExit("--dry-ranks runs in ONE process on one GPU")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        trd = build_trainer(args, device, 1, dry_ranks=args.dry_ranks)
        use_graph = capture_or_eager(trd, args, 0, 1, device)
        w, _ = timed_windows(trd, args.steps if args.steps != 300 else 50, args.warmup, 3, 1, device)
        el = float(np.median(w)); nst = args.steps if args.steps != 300 else 50
        plan = trd.plan_summary()
        print(json.dumps({"metric": f"DRY RUN: rank 0 of a {args.dry_ranks}-rank job on one GPU (B={args.batch_size}/rank, n={args.n}) -- not a multi-GPU measurement",
                          "value": nst / el, "unit": "steps/s (this rank's compute + one-rank collectives)", "ms_per_step": 1e3 * el / nst,
                          "n_gpus": 1, "launch": "hipGraph replay with the RCCL collectives captured" if use_graph else "eager",
                          "final_loss": float(trd.loss_out[3 * trd.B].item()), "plan": plan}))
        dist.destroy_process_group()
        return
    if args.config is not None:        # one conv-config leg on its own (rank 0 of a one-rank run)
        if world != 1:
            raise SystemExit("--config c4|c5 is a one-GPU leg")
        ent = conv_config_leg(args.config, device, steps=args.steps if args.steps != 300 else 20)
        print(json.dumps({"metric": "training steps/sec (BASELINE configs[%d])" % (3 if args.config == "c4" else 4), "n_gpus": 1,
                          "higher_is_better": True, "data": "synthetic", **ent}))
        return
    tr = build_trainer(args, device, world)
    args._f16_headline = bool(getattr(tr, "split_f16", False))
    use_graph = capture_or_eager(tr, args, rank, world, device)
    # "median of 5 windows" -- a short driver run (--steps 20 = ~11 ms of GPU time per window) is then not at the
    # mercy of one scheduling hiccup
    window_s, extra_warm = timed_windows(tr, args.steps, args.warmup, args.windows, world, device)
    elapsed = float(np.median(window_s))
    last = tr.loss_out[3 * tr.B:].clone()
    loss_vals = [float(v) for v in last.cpu()]
    split = bool(tr.split_bf16)
    wide_split = bool(getattr(tr, "split_wgrad_wide", False))
    enc = "10n-50n-50n-50n-50n-10n"

    out = {
        "metric": f"training steps/sec (B={args.batch_size}, n={args.n} MLP)", "value": world * args.steps / elapsed, "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": (("f32 via f16x2 split (3 fp16 MFMA products of scaled two-piece operands, fp32 accumulate)" if getattr(tr, "split_f16", False)
                   else "f32 via bf16x3 split (6 bf16 MFMA products, fp32 accumulate)") if split else
                  ((("f32 via %s split on the 2000-wide layers (forward, data and weight gradients); native fp32 MFMA on the narrow ones"
                     if getattr(tr, "chain", None) else "f32 (forward / data gradients native fp32 MFMA; weight gradients via %s split)")
                    % ("f16x2" if getattr(tr, "split_f16_wide", False) else "bf16x3"))
                   if wide_split else "f32")), "data": "synthetic",
        "global_steps_per_s": args.steps / elapsed,
        "warmup_extra_steps": extra_warm, "windows": len(window_s), "timed_steps_total": args.steps * len(window_s), "timing": "median window of `windows` x `steps` steps",
        "window_ms_per_step": [round(1e3 * w / args.steps, 4) for w in window_s],
        "config": {"workload": f"main_mlp.py --n {args.n} --n-mixing-layer 3 --p {args.p} --batch-size {args.batch_size} "
                               f"--space-type {args.space_type} (unsupervised step: sample->g->f->LpSimCLR->bwd->Adam; encoder {enc})",
                   "batch_per_gpu": args.batch_size, "global_batch": args.batch_size * world,
                   "negatives_pool": args.batch_size * world, "parallelism": f"dp{world}",
                   "launch": "hipGraph replay" if use_graph else "eager"},
        "final_loss": loss_vals[0], "final_pos": loss_vals[1], "final_neg": loss_vals[2],
    }
    out["encoder_arithmetic"] = (
        ("f16x2 split (fp32 emulation, round 5): both fp32 operands of every encoder GEMM -- forward stack, backward data chain AND weight "
         "gradients -- are scaled by a per-tensor power of two and split into two fp16 pieces (hi = RN(v s), lo = RN(v s - hi): 22 "
         "significand bits), the three piece products hi.hi, hi.lo, lo.hi run on the fp16 matrix cores with fp32 accumulation (max error vs "
         "fp64 at the native fp32-MFMA kernels' level, tests/test_gpu_mlp.py); scales follow the previous step's recorded maxima on the device, "
         "overflow raises a sticky flag (`arith_state`); every -m gpu engine test runs in this mode, in bf16x3 and in native fp32 against the "
         "same goldens / tolerances (tests/conftest.py: encoder_arith)") if getattr(tr, "split_f16", False) else
        "split-bf16 (fp32 emulation): both fp32 operands of every encoder GEMM -- forward stack, backward data chain AND weight "
        "gradients -- are split exactly into three bf16 pieces, the six piece products of order <= 2 run on the bf16 matrix cores "
        "with fp32 accumulation (max error vs fp64 8.6e-7 of max|y|, native fp32 MFMA 1.0e-6); every -m gpu engine test runs in "
        "this mode and in native fp32 against the same goldens / tolerances (tests/conftest.py: encoder_arith)"
        if split else (("per-layer kernels (a width beyond 512).  Layers wide on both sides (>= 1024: the 2000 x 2000 ones): forward, data "
                        "gradient and weight gradient in the split-bf16 arithmetic (gemm_split_k / wgrad_split_k on bf16-plane operands, fused "
                        "epilogues); the narrow layers: native fp32 MFMA forward / data gradient, split-bf16 weight gradient on converted planes"
                        if getattr(tr, "chain", None) else
                        "native fp32 MFMA per-layer kernels for the forward and the data gradients (a width beyond 512); weight gradients of the "
                        "MFMA-sized layers in the split-bf16 arithmetic on plane copies made by clica_mlp_planes_from_f32") if wide_split else
                       "native fp32 MFMA" + ("" if tr.fused_forward else " (per-layer kernels: a width beyond 512)")))
    if world > 1:
        comm = comm_leg(tr, rank, world, device)
        if rank == 0:
            out["ranks"] = comm
    if rank == 0 and not args.no_roofline:
        roof, rows = roofline_leg(tr)
        if world == 1 and not args.no_traffic:
            tb_, src_ = measure_traffic(args, roof["kernel"])
            if tb_ is not None:
                roof["traffic_committed_profile"] = roof.get("traffic")
                roof["traffic"], roof["traffic_source"] = tb_, src_
            else:
                roof["traffic_source"] = (roof.get("traffic_source") or "") + f" [in-run measurement unavailable: {src_}]"
        out["roofline"] = roof
        out["kernels"] = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]
        ll = loss_leg(tr)
        out["loss_kernel"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ll.items()}
        if world == 1 and (args.n, args.space_type, args.p) == (10, "box", 2):
            # the same sweeps at the negatives pool of the 8-GPU job (49 152 rows: what bounds weak scaling)
            tr8 = build_trainer(args, device, 1, emulate_pool_ranks=8)
            for _ in range(3):
                tr8.step()
            torch.cuda.synchronize()
            l8 = loss_leg(tr8, reps=10)
            out["loss_kernel_pool_49152"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in l8.items()}
            del tr8
            torch.cuda.empty_cache()
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.torch_port import time_reference_step      # checker/baseline leg only
        threads = torch.get_num_threads()
        med, _ = time_reference_step(n=args.n, B=args.batch_size, p=args.p, steps=5, warmup=2)
        out["cpu_baseline"] = {"value": 1.0 / med, "unit": "steps/s", "cores": threads, "kind": "port",
                               "sample": f"5 timed + 2 warm-up full steps at B={args.batch_size}, n={args.n} (median), "
                                         f"torch {torch.__version__} CPU ops, {os.cpu_count()} host cores visible"}
        out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0 and world == 1 and not args.no_dropin:
        out["dropin"] = dropin_leg(args, device)
    if rank == 0 and world == 1 and split and not args.no_native_leg:
        # the same step on the native fp32-MFMA kernels: its own value, its own roofline entry (fp32 flops / fp32 matrix peak)
        del tr
        torch.cuda.empty_cache()
        tr2 = build_trainer(args, device, world, split_bf16=False)
        capture_or_eager(tr2, args, rank, world, device)
        w2, _ = timed_windows(tr2, min(args.steps, 200), args.warmup, 3, world, device)
        el = float(np.median(w2)); nst = min(args.steps, 200)
        leg = {"value": nst / el, "unit": "steps/s", "ms_per_step": 1e3 * el / nst, "steps": nst, "windows": len(w2), "dtype": "f32",
               "final_loss": float(tr2.loss_out[3 * tr2.B].item()),
               "what": "identical step, encoder GEMMs on v_mfma_f32_16x16x4_f32 / 32x32x2_f32 (split_bf16=False / CLICA_SPLIT_BF16=0)"}
        if not args.no_roofline:
            roof2, _ = roofline_leg(tr2, reps=10)
            leg["roofline"] = {k: roof2[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches_per_step",
                                                     "algorithmic_gflop_per_launch", "dtype")}
        out["native_fp32"] = leg
    if rank == 0 and world == 1 and not args.no_native_leg and getattr(args, "_f16_headline", False):
        # the round-3 / round-4 arithmetic (bf16x3: six bf16 products, 6 B/element planes) next to the f16x2 headline, same box, same call
        tr3 = build_trainer(args, device, world, split_arith="bf16")
        capture_or_eager(tr3, args, rank, world, device)
        w3, _ = timed_windows(tr3, min(args.steps, 200), args.warmup, 3, world, device)
        el = float(np.median(w3)); nst = min(args.steps, 200)
        out["split_bf16x3"] = {"value": nst / el, "unit": "steps/s", "ms_per_step": 1e3 * el / nst, "steps": nst, "windows": len(w3),
                               "final_loss": float(tr3.loss_out[3 * tr3.B].item()),
                               "what": "identical step in the bf16x3 split arithmetic (CLICA_SPLIT_ARITH=bf16: the headline arithmetic of rounds 3-4)"}
        del tr3
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_dry_leg and (args.n, args.space_type, args.p) == (10, "box", 2):
        out["dry_ranks_8"] = dry_ranks_leg(args, device, R=8)
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_secondary and (args.n, args.space_type, args.p) == (10, "box", 2):
        out["secondary"] = secondary_leg(args, device)
        if not args.no_conv_configs:
            out["secondary"]["c4_3dident_resnet18"] = conv_config_leg("c4", device)
            torch.cuda.empty_cache()
            out["secondary"]["c5_kitti_masks"] = conv_config_leg("c5", device)
            try:
                out["secondary"]["c5_kitti_masks"]["roofline"] = c5_conv_roofline(device, traffic=not args.no_traffic)
            except Exception as e:      # noqa: BLE001
                out["secondary"]["c5_kitti_masks"]["roofline"] = {"error": f"{type(e).__name__}: {e}"}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL prints a version banner through C stdio, which is fully buffered on a pipe and would come out at process exit -- BEHIND the
    # contract line.  Flush the C streams first so that the JSON line is the last line of stdout.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
