"""Fused, HBM-resident contrastive train step: the counterpart of ``train_step`` in
/root/reference/main_mlp.py:258-285 plus the sampling at :328, without autograd and without a
host sync.

One step enqueues, on one HIP stream:
    tick -> sample z, z~ (Philox) -> mixing net g -> 7 fused Linear(+LeakyReLU) GEMMs over the
    stacked 2B rows -> [head] -> tiled Lp-InfoNCE forward -> its backward (dz1 += dz3: the
    reference's z3_rec = roll(z1_rec) is, up to a row permutation the row-wise log-sum-exp cannot
    see, "all z1_rec of the batch") -> [head bwd] -> wgrad/dgrad GEMMs into one flat gradient arena
    -> [bucketed RCCL all-reduce] -> one fused Adam launch over the flat parameter arena.
All buffers are preallocated, so the whole step can be captured in a HIP graph (``capture()``).

Data parallel (one process per GPU): every rank samples its own B pairs, all-gathers z1_rec
(B*n*4 bytes per rank) to form the global negatives pool, reduce-scatters d/dz3, and all-reduces the
flat gradient arena; semantics = the single-process loss on the concatenated batch.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import nn

from . import _lib, ops
from . import layers as ls
from contextlib import nullcontext as _nullctx

from .distributed import GradBuckets

__all__ = ["SamplerSpec", "ContrastiveTrainer"]


@dataclass
class SamplerSpec:
    """Ground-truth latent distribution (main_mlp.py:136-200)."""
    space: str = "box"              # box | sphere | real  (--space-type, "unbounded" -> real)
    n: int = 10
    box: tuple = (0.0, 1.0)         # --box-min / --box-max
    marginal: str = "uniform"       # uniform | normal | laplace | gennorm   (--m-p 0/2/1/k)
    m_param: float = 1.0
    m_p: float = 2.0
    conditional: str = "normal"     # normal | laplace | gennorm | vmf        (--c-p 2/1/k/0)
    c_param: float = 0.05
    c_p: float = 2.0
    seed: int = 0


def _abort_capture(graph, stream, others=()) -> None:
    """Leave stream-capture mode after a failed capture, whatever state torch's CUDAGraph object is in (clica_abort_capture
    runs inside the HIP runtime instance torch and the kernels share; a ctypes handle to libamdhip64 may be another copy)."""
    try:
        graph.capture_end()
    except Exception:
        pass
    lib = _lib.load()
    for st in (stream,) + tuple(o for o in others if o is not None):       # the origin stream first, then forked side streams
        lib.clica_abort_capture(C.c_void_p(st.cuda_stream))


class ContrastiveTrainer:
    def __init__(self, f: nn.Sequential, g_weights: torch.Tensor, sampler: SamplerSpec, batch_size: int,
                 p: float = 2, tau: float = 1.0, alpha: float = 0.5, lr: float = 1e-4, g_slope: float = 0.2,
                 betas=(0.9, 0.999), eps: float = 1e-8, device=None,
                 process_group: Optional[dist.ProcessGroup] = None, bucket_bytes: int = 8 << 20,
                 force_collectives: bool = False, overlap_backward: bool = True, fused_forward: bool = True,
                 split_bf16: Optional[bool] = None, g_act_kind: int = 0, emulate_pool_ranks: int = 1, dry_ranks: int = 1,
                 split_arith: Optional[str] = None):
        self.device = torch.device(device if device is not None else "cuda")
        self.f = f.to(self.device)
        self.B = int(batch_size)
        self.n = sampler.n
        self.sampler = sampler
        self.p, self.tau, self.alpha = float(p), float(tau), float(alpha)
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.g_slope = float(g_slope)
        self.g_act_kind = int(g_act_kind)     # hidden activation of g (ops.MIX_ACT); the fused prologue handles LeakyReLU / ReLU only
        self.overlap_backward = bool(overlap_backward)
        self.gW = g_weights.detach().to(self.device, torch.float32).contiguous()
        assert self.gW.shape[1:] == (self.n, self.n)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        # run the collectives even at world size 1 (test hook: exercises the DP code path on one GPU)
        self.force_collectives = bool(force_collectives)
        self.dp = self.world > 1 or (force_collectives and dist.is_initialized())
        # measurement aid (bench.py `secondary` leg): ONE rank's compute of an R-rank data-parallel job on one GPU -- the negatives
        # pool holds R copies of the local embeddings (R x B rows, device-to-device copies stand in for the two all-gathers), so
        # the pair sweeps do the work they do on an R-GPU node.  Not a training mode: every negative counts R times.
        self.emulate_pool = int(emulate_pool_ranks) if (self.world == 1 and not self.dp) else 1
        # DRY RUN of an R-rank job on one GPU (bench.py --dry-ranks R, tests/test_gpu_engine.py): this process plans and runs exactly
        # what rank 0 of an R-rank data-parallel job would -- pool of R x B rows, loss workspaces and stream splits for that pool, the
        # two-half weight-gradient launch, the gradient buckets, 1 / R gradient scale, every collective issued on the (world-1) RCCL
        # group and captured into the step graph -- with the other ranks' contributions stood in for by copies of its own (all-gather
        # results replicated, all-reduce over one rank).  Whatever an 8-GPU run can trip over on the host side (a workspace sized for
        # the wrong plan, a capture RCCL refuses, a bucket boundary off by one) trips here; what it cannot show is the wire.
        self.dry_ranks = int(dry_ranks) if (self.world == 1 and self.dp) else 1
        if int(dry_ranks) > 1 and self.dry_ranks == 1:
            raise ValueError("dry_ranks needs an initialised one-rank process group and force_collectives=True")
        if self.p == 0:
            raise NotImplementedError("p=0 (SimCLRLoss) runs through cl_ica_amd.losses.SimCLRLoss, not the fused engine")

        mods = list(self.f)
        self.linears: List[nn.Linear] = [m for m in mods if isinstance(m, nn.Linear)]
        slopes = {m.negative_slope for m in mods if isinstance(m, nn.LeakyReLU)}
        self.slope = slopes.pop() if slopes else 0.01
        heads = [m for m in mods if isinstance(m, (ls.RescaleLayer, ls.SoftclipLayer))]
        self.head = heads[0] if heads else None
        self._flatten_parameters()
        self.fused_forward = bool(fused_forward) and ops.mlp_fwd_fusable([lin.weight for lin in self.linears])
        # mixing net g inside the fused forward's prologue (needs the one-launch forward, n <= 16)
        self.mix_in_forward = self.fused_forward and self.n <= 16 and self.g_act_kind == 0
        self._x_pending = False
        self.packed = None
        self.packed_t = None
        self._packed_current = False
        self.fused_backward = self.fused_forward and len(self.linears) > 1
        # encoder arithmetic: the whole-stack kernels and the weight gradients on the bf16 matrix cores with exact 3-way bf16
        # operand splits (fp32 emulation: six bf16 products per fp32 product, fp32 accumulate, fp32-grade error) --
        # the default where the encoder fits the whole-stack kernels; split_bf16=False / CLICA_SPLIT_BF16=0 = native fp32 MFMA
        want_split = (os.environ.get("CLICA_SPLIT_BF16", "1") == "1") if split_bf16 is None else bool(split_bf16)
        self._want_split = want_split
        self.split_bf16 = self.fused_backward and want_split and all(lin.bias is not None for lin in self.linears) and \
            sum((lin.out_features + 31) // 32 * 32 for lin in self.linears) <= 3456      # on-chip bias table (fused_mlp.hip)
        # which split arithmetic (round 5): "f16" = two fp16 pieces per operand with per-tensor power-of-two scales, three MFMA products
        # (half the matrix work, 4 B/element; include/clica.h "f16x2 arithmetic") -- the default; "bf16" = the round-3 bf16x3 scheme
        # (six products, 6 B/element, full fp32 exponent range).  CLICA_SPLIT_ARITH / split_arith select; a slope outside (0, 1) keeps bf16.
        arith = (split_arith or os.environ.get("CLICA_SPLIT_ARITH", "f16")).lower()
        if arith not in ("f16", "bf16"):
            raise ValueError(f"split_arith must be 'f16' or 'bf16', got {arith!r}")
        self._arith_f16 = bool(arith == "f16" and 0.0 < self.slope < 1.0 and self.device.type == "cuda")
        self.split_f16 = bool(self.split_bf16 and self._arith_f16)          # whole-stack kernels in f16x2
        self.split_f16_wide = False                                          # per-layer wide path in f16x2 (decided in _allocate)
        self.s16 = None
        self._s16_calibrated = False
        if self.world > 1:
            # identical replicas by construction: rank 0's parameters and mixing weights win (callers that seed every rank
            # identically are unaffected; callers that do not would otherwise train `world` different models silently)
            dist.broadcast(self.param_arena, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                           group=process_group)
            dist.broadcast(self.gW, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                           group=process_group)
        self._allocate()
        if self.split_f16 or self.split_f16_wide:
            self.s16 = ops.Split16(len(self.linears), self.device)
            self._init_wide_ones()
            if self.dp:      # all ranks must take the guard's decision together: the verdict is all-reduced with the gradients
                self.s16.set_dp_poison(self._guard_slot)
        self._watch_versions = True            # step(): parameters written from outside (load_state_dict, copy_) -> recalibrate the f16x2 scales
        self._versions_seen = self._param_versions()
        self._guard_skipped_seen = 0
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        # data parallel + grouped weight gradients: the grouped launch is issued in TWO halves (layers L-1 .. h, then h-1 .. 0) and
        # the first half's slice of the gradient arena is all-reduced on the communication stream while the second half's GEMMs run
        L = len(self.linears)
        self.wgrad_halves = bool(self.dp and self.fused_backward and self.grouped_wgrad and L >= 4)
        self._half = L // 2
        if self.wgrad_halves and self.group_ws is not None:
            # each half's launch plans its own contraction splits (fewer tiles -> more splits per layer -> larger slabs than in the
            # all-layer plan the workspace was sized for: 59.3 MB against 58.5 MB for the n = 10 stack at B = 6144)
            shapes = [tuple(lin.weight.shape) for lin in self.linears]
            mk = ops.mlp_wgrad_split_workspace if self.split_wgrad else ops.mlp_wgrad_workspace
            for part in (shapes[self._half:], shapes[:self._half]):
                w = mk(2 * self.B, part, self.device)
                if w.numel() > self.group_ws.numel():
                    self.group_ws = w
        slices = list(self._layer_slices)
        self._guard_rides = bool(self.dp and self.s16 is not None and self.fused_backward and slices)
        if self._guard_rides:
            # whole-stack path: every producer of the step has run when the first bucket (the last layer's slice: the arena's tail) goes
            # out, so the verdict slot behind the arena rides in that bucket; the per-layer path reduces the slot on its own at the end
            total = self.grad_arena.numel()
            k = max(range(len(slices)), key=lambda i: slices[i][1])
            if k == 0 and slices[k][1] >= total - 3:
                slices[k] = (slices[k][0], total + 4)
            else:
                self._guard_rides = False
        self.buckets = GradBuckets(self._grad_buf if self._guard_rides else self.grad_arena, slices, self.world, process_group, bucket_bytes,
                                   force=self.dp, boundaries=(L - 1 - self._half,) if self.wgrad_halves else ()) if self.dp else None

    # -------------------------------------------------------------------------------- arenas
    def _flatten_parameters(self):
        """Re-point every parameter of f into one 16-byte aligned flat arena (Parameter objects and
        state-dict keys are unchanged) and create matching grad / exp_avg / exp_avg_sq arenas."""
        params = list(self.f.parameters())
        offs, total = [], 0
        for prm in params:
            offs.append(total)
            total += (prm.numel() + 3) // 4 * 4
        dev = self.device
        self.param_arena = torch.zeros(total, dtype=torch.float32, device=dev)
        # (+ 4 floats behind the gradients: the f16x2 guard's verdict slot, which rides through the data-parallel all-reduce of the
        #  last layer's bucket -- include/clica.h, "THE GUARD of the f16x2 arithmetic")
        self._grad_buf = torch.zeros(total + 4, dtype=torch.float32, device=dev)
        self.grad_arena = self._grad_buf[:total]
        self._guard_slot = self._grad_buf[total:total + 1]
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._views, self._gviews = {}, {}
        for prm, off in zip(params, offs):
            view = self.param_arena[off:off + prm.numel()].view(prm.shape)
            view.copy_(prm.data.to(dev))
            prm.data = view
            gview = self.grad_arena[off:off + prm.numel()].view(prm.shape)
            prm.grad = gview
            self._views[id(prm)] = view
            self._gviews[id(prm)] = gview
        # slices per Linear layer in BACKWARD completion order (last layer first) for bucketing
        self._layer_slices = []
        pid = {id(prm): (off, prm.numel()) for prm, off in zip(params, offs)}
        for lin in reversed(self.linears):
            o_w, n_w = pid[id(lin.weight)]
            o_b, n_b = pid[id(lin.bias)]
            self._layer_slices.append((min(o_w, o_b), max(o_w + n_w, o_b + n_b)))
        # parameters that are not Linear weights/biases (a learnable head: RescaleLayer.r / SoftclipLayer.max_abs_bound) get their
        # gradient at the very start of backward(): they ride in the first slice to complete, so the data-parallel all-reduce
        # covers them too (otherwise every rank would train its own head parameter)
        lin_ids = {id(q) for lin in self.linears for q in (lin.weight, lin.bias)}
        for prm in params:
            if id(prm) not in lin_ids and self._layer_slices:
                o, k = pid[id(prm)]
                lo, hi = self._layer_slices[0]
                self._layer_slices[0] = (min(lo, o), max(hi, o + (k + 3) // 4 * 4))
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)

    def _allocate(self):
        dev, B, n = self.device, self.B, self.n
        R = 2 * B
        f32 = dict(dtype=torch.float32, device=dev)
        self.z = torch.empty((R, n), **f32)            # rows [0,B): z ; [B,2B): z~
        self.x = torch.empty((R, n), **f32)            # g(z), g(z~)
        widths = [lin.out_features for lin in self.linears]
        self.acts = [torch.empty((R, w), **f32) for w in widths]     # post-activation outputs; last = pre-head
        self.y = torch.empty((R, n), **f32) if self.head is not None else self.acts[-1]
        self.inv_norm = torch.empty((R,), **f32) if isinstance(self.head, ls.RescaleLayer) else None
        wmax = max(widths + [n])
        self.dbuf = [torch.empty((R, wmax), **f32) for _ in range(3)]
        self.side_stream = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and self.overlap_backward) else None
        self.dy = torch.empty((R, n), **f32)
        self.loss_out = torch.empty(3 * B + 3, **f32)
        Bg = B * self.world * self.emulate_pool * self.dry_ranks
        pooled = self.dp or self.emulate_pool > 1
        self.z_all = torch.empty((Bg, n), **f32) if pooled else None
        self.lse_all = torch.empty((Bg,), **f32) if pooled else None
        self.desc = _lib.LpLossDesc(B=B, B3=Bg, n=n, p=self.p, tau=self.tau, alpha=self.alpha, compat=1, pow=1)
        fb, bb = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().clica_lp_loss_workspace_bytes(C.byref(self.desc), C.byref(fb), C.byref(bb)), "workspace")
        tb = C.c_size_t()
        self.loss_train = self.p >= 1                  # fused training pair of loss entry points
        if self.loss_train:
            _lib.check(_lib.load().clica_lp_loss_train_workspace_bytes(C.byref(self.desc), C.byref(tb)), "train workspace")
        self.loss_ws = torch.zeros(max(fb.value, bb.value, tb.value), dtype=torch.uint8, device=dev)
        # the step / RNG counter is advanced by the loss backward's reduction launch (all samplers of the step have run by then)
        # instead of a separate one-thread launch at the end; Adam then takes t = counter
        self.early_tick = self.loss_train
        self._ticked = False
        nb = C.c_size_t(); need = 0
        for lin in self.linears:
            _lib.check(_lib.load().clica_linear_wgrad_workspace_bytes(R, lin.out_features, lin.in_features, C.byref(nb)), "wgrad ws")
            need = max(need, nb.value)
        self.wgrad_ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        self.wgrad_ws2 = torch.zeros(need, dtype=torch.uint8, device=dev)    # second stream's slabs
        self.dz = [torch.empty((R, w), **f32) for w in widths[:-1]] if self.fused_backward else None   # dZ_l for the wgrads
        self.grouped_wgrad = self.fused_backward and all(lin.bias is not None for lin in self.linears)
        self.group_ws = ops.mlp_wgrad_workspace(R, [tuple(lin.weight.shape) for lin in self.linears], dev) if self.grouped_wgrad else None
        # sign bits of every hidden activation, written by the fused forward, read by the fused backward chain
        self.signmasks = (ops.mlp_signmask_alloc(R, len(self.linears) - 1, dev) + [None]) if self.fused_backward else None
        # split-bf16 weight gradients (csrc/wgrad_split.hip): the MFMA-sized layers read BOTH operands as bf16 planes that the
        # split forward / backward-chain kernels write instead of the fp32 copies (nobody else reads a hidden activation or
        # dZ); the tiny first / last layer keeps the fp32 VALU kernel, so the tensors next to them stay fp32 as well.
        L = len(self.linears)
        kinds = [ops.mlp_wgrad_split_kind(lin.out_features, lin.in_features) for lin in self.linears] if self.split_bf16 else []
        self.split_wgrad = bool(self.split_bf16 and self.grouped_wgrad and L >= 3 and kinds[0] == 1 and kinds[-1] == 1)
        self.act_planes = [None] * L
        self.dz_planes = [None] * L
        self.acts_out = list(self.acts)                      # what the forward writes as fp32 (None: planes only)
        self.dz_out = list(self.dz) if self.dz is not None else None
        if self.split_wgrad:
            keep = bool(getattr(self, "keep_fp32_copies", False))          # (inspection: also write every fp32 copy)
            for l in range(L):
                if kinds[l] == 0:                            # dW_l = dZ_l^T acts_{l-1} on the bf16 / fp16 matrix cores
                    self.act_planes[l - 1] = ops.mlp_planes_alloc(R, widths[l - 1], True, dev, f16=self.split_f16)
                    self.dz_planes[l] = ops.mlp_planes_alloc(R, widths[l], False, dev, f16=self.split_f16)
            for l in range(L - 1):
                if not keep and kinds[l + 1] == 0:           # acts[l] feeds only an MFMA-sized weight gradient
                    self.acts_out[l] = None
                if not keep and kinds[l] == 0:
                    self.dz_out[l] = None
            self.group_ws = ops.mlp_wgrad_split_workspace(R, [tuple(lin.weight.shape) for lin in self.linears], dev)
        # Wide encoders (a width beyond 512: the per-layer fp32 GEMM path, BASELINE config 3): forward and data gradients stay on
        # the fp32-MFMA kernels, the WEIGHT gradients of the MFMA-sized layers run in the split-bf16 arithmetic on plane copies
        # that an HBM-bound conversion kernel makes of the fp32 activations / gradients (clica_mlp_planes_from_f32)
        self.split_wgrad_wide = bool(self._want_split and not self.fused_forward and not self.fused_backward
                                     and all(lin.bias is not None for lin in self.linears))
        if self.split_wgrad_wide:
            # (round 5) the per-layer split kernels in the f16x2 arithmetic too: plane copies of 4 B/element, three products, per-tensor scales
            # in the same Split16 state (tensor = (family, layer): activations by the layer they feed, gradients and weights by their layer)
            self.split_f16_wide = f16w = bool(self._arith_f16 and L <= 8)
            self.wide_kinds = [ops.mlp_wgrad_split_kind(lin.out_features, lin.in_features) for lin in self.linears]
            in_w = [lin.in_features for lin in self.linears]
            self.xin_planes = [ops.mlp_planes_alloc(R, in_w[l], True, dev, f16=f16w) if self.wide_kinds[l] == 0 else None for l in range(L)]
            self.dzw_planes = [ops.mlp_planes_alloc(R, widths[l], False, dev, f16=f16w) if self.wide_kinds[l] == 0 else None for l in range(L)]
            shapes = [tuple(lin.weight.shape) for lin in self.linears]
            self.wide_ws = max((ops.mlp_wgrad_split_workspace(R, [shapes[l]], dev) for l in range(L) if self.wide_kinds[l] == 0),
                               key=lambda t: t.numel(), default=None)
        # ... and, for the layers that are wide on BOTH sides (>= 1024: config 3's three 2000 x 2000 layers), forward and data
        # gradient too: split-bf16 GEMMs with fused epilogues on T-plane operands (csrc/wgrad_split.hip: gemm_split_k; 1.7x the
        # fp32-MFMA kernels at 12288 x 2000 x 2000).  A chain layer reads its input as T-planes (planes of the transposed
        # activation), written by the previous chain layer's epilogue or converted from fp32 at the chain's head, and writes
        # T-planes for the next chain layer, N-planes for the next layer's weight gradient, fp32 only where an fp32 kernel follows.
        self.chain = set()
        if self.split_wgrad_wide:
            in_w = [lin.in_features for lin in self.linears]
            # (bf16x3: 1024 -- with the 400-wide layers in the chain one p = 1 gradient check of G13 measures 1.1e-5; f16x2: 384 -- all of
            #  config 3's checks hold 1e-5 and the step gains 13 %: 198 -> 225 steps/s)
            cmin = 384 if self.split_f16_wide else 1024
            self.chain = {l for l in range(1, L - 1) if self.wide_kinds[l] == 0 and in_w[l] >= cmin and widths[l] >= cmin}
        if self.chain:
            in_w = [lin.in_features for lin in self.linears]
            f16w = self.split_f16_wide
            self.wT = {l: ops.mlp_planes_alloc(in_w[l], widths[l], False, dev, f16=f16w) for l in self.chain}      # planes of W^T
            self.wN = {l: ops.mlp_planes_alloc(widths[l], in_w[l], False, dev, f16=f16w) for l in self.chain}      # planes of W
            self.xT = {l: ops.mlp_planes_alloc(in_w[l], R, False, dev, f16=f16w) for l in self.chain}              # T-planes of the layer input
            self.dzT = {l: ops.mlp_planes_alloc(widths[l], R, False, dev, f16=f16w) for l in self.chain}           # T-planes of dZ_l
            if not f16w:
                self._init_wide_ones()
            self._wide_packed = False
        if self.head is not None:
            self.head_part = torch.empty(((R + 255) // 256, n if isinstance(self.head, ls.SoftclipLayer) else 1), **f32)
            hp = self.head.r if isinstance(self.head, ls.RescaleLayer) else self.head.max_abs_bound
            self.head_param = hp.data if isinstance(hp, nn.Parameter) else hp.to(dev).contiguous()
            self.head_learnable = isinstance(hp, nn.Parameter)
            self.dpre = torch.empty((R, n), **f32)

    def _init_wide_ones(self):
        """The fused epilogues of the wide chain never touch the constant-1 column of the N-planes they fill: written once here."""
        if not getattr(self, "chain", None):
            return
        L, R = len(self.linears), 2 * self.B
        widths = [lin.out_features for lin in self.linears]
        for l in self.chain:
            if l + 1 < L and self.wide_kinds[l + 1] == 0:
                ops.mlp_planes_from_f32(torch.zeros((R, widths[l]), dtype=torch.float32, device=self.device), True, out=self.xin_planes[l + 1],
                                        **self._s16w(0, l + 1))

    def _s16w(self, family: int, index: int) -> dict:
        """Keyword arguments that name a tensor of the f16x2 state for the per-layer producers (empty in the bf16x3 arithmetic)."""
        return dict(state=self.s16, tensor=(family, index)) if self.split_f16_wide else {}

    # -------------------------------------------------------------------------------- data
    def sample(self):
        """z ~ marginal, z~ ~ conditional(z) on device (main_mlp.py:196-200), then x = g(z)."""
        s, B, n = self.sampler, self.B, self.n
        z, zt = self.z[:B], self.z[B:]
        sid = 2 * self.rank
        mean = None
        if s.marginal != "uniform":
            if not hasattr(self, "_eta"):
                self._eta = torch.zeros(1, n, device=self.device)
                if s.space == "sphere":
                    self._eta[0, 0] = 1.0                      # main_mlp.py:148-150
            mean = self._eta
        # both draws in one launch for the coordinate-wise kinds (box, R^n); two for the sphere / vMF
        ops.sample_pair(s.space, s.marginal, s.conditional, n, B, z, zt, marginal_mean=mean, m_scale=s.m_param, m_p=s.m_p,
                        c_scale=s.c_param, c_p=s.c_p, box=s.box, seed=s.seed, stream_id=sid, step_dev=self.step_dev)
        self._mix()

    def inject(self, z1: torch.Tensor, z2: torch.Tensor):
        """Use caller-provided latents instead of the device sampler (parity tests)."""
        self.z[:self.B].copy_(z1); self.z[self.B:].copy_(z2)
        self._mix()

    def _mix(self):
        """x = g(z): its own launch, unless the fused forward runs the mixing net in its prologue."""
        self._x_pending = self.mix_in_forward
        if not self.mix_in_forward:
            ops.mixing_fwd(self.z, self.gW, self.g_slope, out=self.x, act_kind=self.g_act_kind)

    # -------------------------------------------------------------------------------- step pieces
    def _pack_sample_merged(self) -> bool:
        """The step's two independent front launches -- weight pack and latent pair draw -- in ONE (clica_mlp_pack_split16_both_sample):
        f16x2 arithmetic, packed buffers already allocated, the one-launch sampler path.  False: the caller runs them separately."""
        if self.s16 is None or not self.split_bf16 or self.packed is None or self.packed_t is None:
            return False
        s, B, n = self.sampler, self.B, self.n
        mean = None
        if s.marginal != "uniform":
            if not hasattr(self, "_eta"):
                self._eta = torch.zeros(1, n, device=self.device)
                if s.space == "sphere":
                    self._eta[0, 0] = 1.0                      # main_mlp.py:148-150
            mean = self._eta
        ops.mlp_pack_split16_sample([lin.weight for lin in self.linears], self.packed, self.packed_t, self.s16, s.space, s.marginal,
                                    s.conditional, n, B, self.z[:B], self.z[B:], marginal_mean=mean, m_scale=s.m_param, m_p=s.m_p,
                                    c_scale=s.c_param, c_p=s.c_p, box=s.box, seed=s.seed, stream_id=2 * self.rank, step_dev=self.step_dev)
        self._packed_current = True
        self._mix()
        return True

    def pack(self):
        """Fragment-order copies of the CURRENT weights for the fused forward / backward-chain kernels (one launch
        for both layouts when both are used).  Valid until the next optimizer step."""
        ws = [lin.weight for lin in self.linears]
        if self.split_bf16:
            self.packed, self.packed_t = ops.mlp_pack_split_both(ws, self.packed, self.packed_t, state=self.s16)
        elif self.fused_backward:
            self.packed, self.packed_t = ops.mlp_pack_both(ws, self.packed, self.packed_t)
        elif self.fused_forward:
            self.packed = ops.mlp_pack_weights(ws, self.packed)
        self._packed_current = True

    def forward(self):
        cur = self.x
        L = len(self.linears)
        self._planes_on_current_scales = True      # (saved_activation: the plane copies this pass writes are scaled by the scales in force NOW)
        if self.fused_forward:
            # one launch for the whole stack, activation panel resident in LDS (csrc/fused_mlp.hip)
            ws = [lin.weight for lin in self.linears]
            if not self._packed_current:
                self.pack()
            mix = None
            if self._x_pending:          # latents in, x = g(z) computed in the kernel prologue and stored to self.x
                cur, mix, self._x_pending = self.z, (self.gW, self.g_slope, self.x), False
            if self.split_bf16:
                ops.mlp_fwd_split(cur, ws, [lin.bias for lin in self.linears], self.acts_out, self.packed, self.slope,
                                  signmasks=self.signmasks, mix=mix, planes=self.act_planes if self.split_wgrad else None, state=self.s16)
            else:
                ops.mlp_fwd(cur, ws, [lin.bias for lin in self.linears], self.acts, self.slope, packed=self.packed,
                            signmasks=self.signmasks, mix=mix)
            cur = self.acts[-1]
        else:
            wide = getattr(self, "split_wgrad_wide", False)
            chain = getattr(self, "chain", set())
            if chain and not (self._wide_packed and self._packed_current):
                for l in chain:                             # plane copies of the CURRENT weights, both orientations
                    ops.mlp_planes_from_f32_t(self.linears[l].weight, out=self.wT[l], **self._s16w(2, l))
                    ops.mlp_planes_from_f32(self.linears[l].weight, False, out=self.wN[l], **self._s16w(2, l))
                self._wide_packed = self._packed_current = True
            R = self.x.shape[0]
            for l, lin in enumerate(self.linears):
                fed = (l - 1) in chain                      # the previous layer's epilogue already wrote this layer's plane operands
                if wide and self.wide_kinds[l] == 0 and not fed:        # this layer's input as bf16 planes for its weight gradient
                    ops.mlp_planes_from_f32(cur, True, out=self.xin_planes[l], **self._s16w(0, l))
                if l in chain:
                    if not fed:
                        ops.mlp_planes_from_f32_t(cur, out=self.xT[l], **self._s16w(0, l))
                    nxt = (l + 1) in chain
                    out = None if nxt else self.acts[l]
                    ops.linear_split_fwd(self.xT[l], self.wT[l], lin.bias, R, lin.out_features, lin.in_features, l < L - 1, self.slope,
                                         yT=self.xT[l + 1] if nxt else None,
                                         yN=self.xin_planes[l + 1] if (l + 1 < L and self.wide_kinds[l + 1] == 0) else None, yN_ones=True, y=out,
                                         **(dict(state=self.s16, layer=l) if self.split_f16_wide else {}))
                    cur = out
                    continue
                ops.linear_fwd(cur, lin.weight, lin.bias, leaky=(l < L - 1), slope=self.slope, out=self.acts[l])
                cur = self.acts[l]
        if self.head is not None:
            lib, st = _lib.load(), _lib.stream_ptr()
            R, n = cur.shape
            if isinstance(self.head, ls.RescaleLayer):
                _lib.check(lib.clica_rescale_fwd(cur.data_ptr(), n, self.head_param.data_ptr(), self.y.data_ptr(), n,
                                                 self.inv_norm.data_ptr(), R, n, st), "clica_rescale_fwd")
            else:
                _lib.check(lib.clica_softclip_fwd(cur.data_ptr(), n, self.head_param.data_ptr(), self.y.data_ptr(), n, R, n, st),
                           "clica_softclip_fwd")

    def loss_forward_backward(self):
        """Loss forward + its backward w.r.t. the embeddings.  The negatives are "all z1_rec of the (global)
        batch" (the reference's roll, main_mlp.py:272, up to a permutation), so the pair matrix is symmetric
        and ONE backward sweep with both rows' softmax statistics gives dz1 complete: no row pass, no column
        pass, and under data parallelism no reduce-scatter of d/dz3 -- only an all-gather of the B
        log-sum-exp values next to the all-gather of the embeddings."""
        lib, st = _lib.load(), _lib.stream_ptr()
        B, n, o = self.B, self.n, self.loss_out
        y1, y2 = self.y[:B], self.y[B:]
        lse = o[2 * B:3 * B]
        emu = self.emulate_pool > 1
        dry = self.dry_ranks > 1
        if self.dp:
            dist.all_gather_into_tensor(self.z_all[:B * self.world] if dry else self.z_all, y1.contiguous(), group=self.pg)
            if dry:      # the other ranks' rows: copies of this rank's
                self.z_all.view(self.dry_ranks, B, n)[1:].copy_(self.z_all[:B].unsqueeze(0))
            pool = self.z_all
        elif emu:
            self.z_all.view(self.emulate_pool, B, n).copy_(y1.unsqueeze(0))
            pool = self.z_all
        else:
            pool = y1
        if self.loss_train:
            # forward + coefficient step (finalize), then -- after the all-gather of the row statistics under DP -- pair sweep +
            # reduction (+ the forward's means): 5 launches instead of 8
            _lib.check(lib.clica_lp_loss_fwd_train(C.byref(self.desc), y1.data_ptr(), n, y2.data_ptr(), n, pool.data_ptr(), n,
                                                   o[:B].data_ptr(), o[B:2 * B].data_ptr(), lse.data_ptr(),
                                                   self.dy[:B].data_ptr(), n, self.dy[B:].data_ptr(), n,
                                                   self.loss_ws.data_ptr(), self.loss_ws.numel(), st), "clica_lp_loss_fwd_train")
            if self.dp:
                dist.all_gather_into_tensor(self.lse_all[:B * self.world] if dry else self.lse_all, lse, group=self.pg)
                if dry:
                    self.lse_all.view(self.dry_ranks, B)[1:].copy_(self.lse_all[:B].unsqueeze(0))
            elif emu:
                self.lse_all.view(self.emulate_pool, B).copy_(lse.unsqueeze(0))
            pool_lse = self.lse_all if (self.dp or emu) else lse
            self._dy_parts = None
            if self._chain_takes_partials(emu):
                # N = 1 training step: no reduction launch -- the backward chain's prologue sums the pair sweep's partials into dy, leaves
                # the forward's means and ticks the counter (clica_lp_dy_parts; same sums in the same order)
                self._dy_parts = _lib.DyParts()
                self._dy_parts_taken = getattr(self, "_dy_parts_taken", 0) + 1      # (tests: which path a trainer took)
                _lib.check(lib.clica_lp_loss_bwd_sym_train_parts(C.byref(self.desc), y1.data_ptr(), n, pool.data_ptr(), n,
                                                                 lse.data_ptr(), pool_lse.data_ptr(), o[3 * B:].data_ptr(),
                                                                 self.step_dev.data_ptr() if self.early_tick else None,
                                                                 self.loss_ws.data_ptr(), self.loss_ws.numel(), C.byref(self._dy_parts), st),
                           "clica_lp_loss_bwd_sym_train_parts")
                self._ticked = self.early_tick
                return
            _lib.check(lib.clica_lp_loss_bwd_sym_train(C.byref(self.desc), y1.data_ptr(), n, pool.data_ptr(), n,
                                                       lse.data_ptr(), pool_lse.data_ptr(), self.dy[:B].data_ptr(), n, o[3 * B:].data_ptr(),
                                                       self.step_dev.data_ptr() if self.early_tick else None,
                                                       self.loss_ws.data_ptr(), self.loss_ws.numel(), st), "clica_lp_loss_bwd_sym_train")
            self._ticked = self.early_tick
            return
        _lib.check(lib.clica_lp_loss_fwd(C.byref(self.desc), y1.data_ptr(), n, y2.data_ptr(), n, pool.data_ptr(), n,
                                         o[:B].data_ptr(), o[B:2 * B].data_ptr(), lse.data_ptr(), o[3 * B:].data_ptr(),
                                         None, 0, self.loss_ws.data_ptr(), self.loss_ws.numel(), st), "clica_lp_loss_fwd")
        if self.dp:
            dist.all_gather_into_tensor(self.lse_all[:B * self.world] if dry else self.lse_all, lse, group=self.pg)
            if dry:
                self.lse_all.view(self.dry_ranks, B)[1:].copy_(self.lse_all[:B].unsqueeze(0))
            pool_lse = self.lse_all
        elif emu:
            self.lse_all.view(self.emulate_pool, B).copy_(lse.unsqueeze(0))
            pool_lse = self.lse_all
        else:
            pool_lse = lse
        _lib.check(lib.clica_lp_loss_bwd_sym(C.byref(self.desc), y1.data_ptr(), n, y2.data_ptr(), n, pool.data_ptr(), n,
                                             lse.data_ptr(), pool_lse.data_ptr(), None, None, None,
                                             self.dy[:B].data_ptr(), n, self.dy[B:].data_ptr(), n,
                                             self.loss_ws.data_ptr(), self.loss_ws.numel(), st), "clica_lp_loss_bwd_sym")

    # Test hooks (class attributes, tests/test_gpu_engine.py): both the folded and the unfolded launch structures are product paths -- the
    # unfolded ones run under data parallelism and behind a head -- and the equivalence tests compare them on ONE trainer configuration.
    chain_tail = True            # the backward chain's tail writes the n-wide layers' weight-gradient slabs (clica_mlp_dgrad_split_tail)
    fold_dy_reduce = True        # ... and its prologue finishes dy from the loss sweep's partials (no reduction launch)

    def _chain_tail_ok(self) -> bool:
        if not hasattr(self, "_tail_ok"):
            self._tail_ok = (self.chain_tail and self.dz_out[0] is not None and
                             ops.mlp_chain_tail_supported([tuple(lin.weight.shape) for lin in self.linears]))
        return self._tail_ok

    def _chain_takes_partials(self, emu: bool) -> bool:
        """Will THIS step's backward chain run with its tail (clica_mlp_dgrad_split_tail) directly on dy?  Then it can finish dy itself."""
        if not self.fold_dy_reduce or not getattr(self, "_in_step", False):
            return False
        if self.dp or emu or self.head is not None or not (self.split_bf16 and self.split_wgrad and self.fused_backward):
            return False
        return self._adam_folds_into_wgrad() and self._chain_tail_ok()

    def _head_backward(self):
        """d loss / d (pre-head output): the head's backward (and its parameter gradient), or dy itself."""
        lib, st = _lib.load(), _lib.stream_ptr()
        g = self.dy
        R, n = g.shape
        if self.head is not None:
            pre = self.acts[-1]
            part = self.head_part if self.head_learnable else None
            if isinstance(self.head, ls.RescaleLayer):
                _lib.check(lib.clica_rescale_bwd(pre.data_ptr(), n, self.head_param.data_ptr(), self.inv_norm.data_ptr(),
                                                 g.data_ptr(), n, self.dpre.data_ptr(), n, _lib.ptr(part), R, n, st), "clica_rescale_bwd")
            else:
                _lib.check(lib.clica_softclip_bwd(pre.data_ptr(), n, self.head_param.data_ptr(), g.data_ptr(), n,
                                                  self.dpre.data_ptr(), n, _lib.ptr(part), R, n, st), "clica_softclip_bwd")
            if self.head_learnable:
                hp = self.head.r if isinstance(self.head, ls.RescaleLayer) else self.head.max_abs_bound
                torch.sum(self.head_part, dim=0, out=self._gviews[id(hp)])
            g = self.dpre
        return g

    def backward_chain(self, g):
        """Whole data-gradient chain dZ_{L-1} -> ... -> dZ_0 in ONE launch (dZ panel resident in LDS, transposed
        fragment-order weights); every dZ_l is also written to HBM (fp32 and / or bf16 planes) for the weight gradients."""
        L = len(self.linears)
        chain = list(range(L - 1, 0, -1))
        ws = [self.linears[l].weight for l in chain]
        if not self._packed_current:
            self.pack()
        if self.split_bf16:
            tail = None
            self._tail_ready = False
            if getattr(self, "_fold_adam", False) and self.split_wgrad:
                # inside a training step (N = 1): the chain's workgroups also leave the n-wide first / last layer's weight-gradient
                # slabs (clica_mlp_dgrad_split_tail) -- weight_grads() then needs no tiny-dimension launch
                if self._chain_tail_ok():
                    tail = dict(a_last=self.acts_out[L - 2], x=self.x, shapes=[tuple(lin.weight.shape) for lin in self.linears], ws=self.group_ws,
                                dy_parts=getattr(self, "_dy_parts", None))
                    self._tail_ready = True
            if getattr(self, "_dy_parts", None) is not None and (tail is None or g is not self.dy):
                raise _lib.ClicaError("engine: the loss left its partials for the backward chain, but the chain does not run with its tail on dy")
            self._dy_parts = None
            ops.mlp_dgrad_chain_split(g, ws, self.packed_t, [self.dz_out[l - 1] for l in chain], self.slope,
                                      masks_chain=[self.signmasks[l - 1] for l in chain],
                                      planes=[self.dz_planes[l - 1] for l in chain] if self.split_wgrad else None, state=self.s16, tail=tail)
        else:
            ops.mlp_dgrad_chain(g, ws, self.packed_t, [self.acts[l - 1] for l in chain], [self.dz[l - 1] for l in chain], self.slope,
                                masks_chain=[self.signmasks[l - 1] for l in chain])

    def weight_grads(self, g, layers=None):
        """dW / db of `layers` (default: every layer): tiny-layer launch + one grouped split-K GEMM of equal-length work items +
        one grouped slab reduction (fp32 MFMA on the fp32 copies, or bf16x3 MFMA on the plane copies in split mode)."""
        L = len(self.linears)
        R = g.shape[0]
        order = list(range(L)) if layers is None else list(layers)
        dWs = [self._gviews[id(self.linears[l].weight)] for l in order]
        dbs = [self._gviews[id(self.linears[l].bias)] for l in order]
        if self.split_wgrad:
            adam = None
            if getattr(self, "_fold_adam", False) and layers is None:
                # inside a training step (N = 1): the reduction launch that ends the weight gradients applies the optimizer as well
                # (clica_mlp_wgrad_split_adam) -- optimizer_step() then has nothing left to launch
                adam = dict(param=self.param_arena, grad=self.grad_arena, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq,
                            step_dev=self.step_dev, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                            grad_scale=1.0, t_offset=0 if self._ticked else 1, s16=self.s16)
            ops.mlp_wgrad_split(R, [self.dz_planes[l] for l in order], [self.act_planes[l - 1] if l > 0 else None for l in order],
                                [g if l == L - 1 else self.dz_out[l] for l in order],
                                [self.acts_out[l - 1] if l > 0 else self.x for l in order], dWs, dbs, ws=self.group_ws,
                                state=self.s16, a_index=order, d_index=[L - 1 - l for l in order], adam=adam,
                                tail_slabs=adam is not None and getattr(self, "_tail_ready", False))
            self._tail_ready = False
            if adam is not None:
                self._adam_done = True
        else:
            ops.mlp_wgrad([g if l == L - 1 else self.dz[l] for l in order],
                          [self.acts[l - 1] if l > 0 else self.x for l in order], dWs, dbs, ws=self.group_ws)

    def saved_activation(self, l):
        """fp32 values of layer l's output as saved for the backward pass.  Where a split-bf16 kernel wrote the activation only as
        bf16 planes (no fp32 copy: nobody but the matrix cores reads it) the planes are decoded -- exactly, hi + mid + lo is the
        fp32 value.  Inspection / tests."""
        R, w = self.x.shape[0], self.linears[l].out_features
        # f16x2: the planes are scaled by the activation scale the pass that WROTE them ran with -- the scales in force if no update has run
        # since (a bare forward(), a calibration pass), the previous ones (kept by the update) after a full step
        which = "scales_a" if getattr(self, "_planes_on_current_scales", False) else "last_scales_a"
        if l in getattr(self, "chain", set()) and (l + 1) in self.chain:
            if self.split_f16_wide:
                return ops.mlp_planes16_to_f32(self.xin_planes[l + 1], R, w, True, self.s16.read()[which][l + 1])
            return ops.mlp_planes_to_f32(self.xin_planes[l + 1], R, w, True)
        if self.acts_out[l] is None:
            if self.split_f16:
                return ops.mlp_planes16_to_f32(self.act_planes[l], R, w, True, self.s16.read()[which][l + 1])
            return ops.mlp_planes_to_f32(self.act_planes[l], R, w, True)
        return self.acts[l]

    def _wgrad_layer(self, l, g, inp, ws, planes_ready=False):
        """dW_l / db_l of one layer on the per-layer path: fp32 MFMA GEMM, or -- wide encoders in split mode -- the split-bf16
        kernel on plane copies (dZ converted here, the layer input converted in forward())."""
        lin = self.linears[l]
        dW, db = self._gviews[id(lin.weight)], self._gviews[id(lin.bias)]
        if getattr(self, "split_wgrad_wide", False) and self.wide_kinds[l] == 0:
            if not planes_ready:                            # (a split data-gradient epilogue has written dZ_l's planes already)
                ops.mlp_planes_from_f32(g, False, out=self.dzw_planes[l], **self._s16w(1, l))
            ops.mlp_wgrad_split(self.x.shape[0], [self.dzw_planes[l]], [self.xin_planes[l]], [None], [None], [dW], [db], ws=self.wide_ws,
                                **(dict(state=self.s16, a_index=[l], d_index=[l]) if self.split_f16_wide else {}))
        else:
            ops.linear_wgrad(g, inp, dW=dW, db=db, accumulate=False, ws=ws)

    def backward(self):
        g = self._head_backward()
        L = len(self.linears)
        main = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        side = self.side_stream
        two = side is not None and main is not None
        if self.fused_backward:
            # (1) the data-gradient chain in one launch; (2) the weight-gradient GEMMs
            st = getattr(self, "stamps", None)
            if st:
                ops.stamp(st["mlp_dgrad"], 0)
            self.backward_chain(g)
            if self._guard_rides:                      # this rank's verdict into the slot the first gradient bucket carries
                self.s16.poison_export(self._guard_slot)
            if st:
                ops.stamp(st["mlp_dgrad"], 1)
                if self.grouped_wgrad and self.buckets is None:
                    ops.stamp(st["mlp_wgrad"], 0)
                    self.weight_grads(g)
                    ops.stamp(st["mlp_wgrad"], 1)
                    return
            if self.grouped_wgrad:
                if self.buckets is not None and self.wgrad_halves:
                    h = self._half
                    self.weight_grads(g, layers=range(h, L))          # the head parameter's slot rides in the first slice
                    for i in range(0, L - h):
                        self.buckets.layer_done(i)                    # completion index i = layer L - 1 - i: all-reduce starts now ...
                    self.weight_grads(g, layers=range(0, h))          # ... and runs under this launch
                    for i in range(L - h, L):
                        self.buckets.layer_done(i)
                    self.buckets.wait()
                    return
                self.weight_grads(g)
                if self.buckets is not None:
                    for i in range(L):
                        self.buckets.layer_done(i)
                    self.buckets.wait()
                return
            use_two = two and self.buckets is None
            if use_two:
                side.wait_stream(main)
            for i, l in enumerate(reversed(range(L))):
                lin = self.linears[l]
                gl = g if l == L - 1 else self.dz[l]
                inp = self.acts[l - 1] if l > 0 else self.x
                on_side = use_two and (i & 1)
                with torch.cuda.stream(side) if on_side else _nullctx():
                    ops.linear_wgrad(gl, inp, dW=self._gviews[id(lin.weight)], db=self._gviews[id(lin.bias)], accumulate=False,
                                     ws=self.wgrad_ws2 if on_side else self.wgrad_ws)
                if self.buckets is not None:
                    self.buckets.layer_done(L - 1 - l)
            if use_two:
                main.wait_stream(side)
            if self.buckets is not None:
                self.buckets.wait()
            return
        # Per-layer path (wide encoders).  Two streams: wgrad_l (weight/bias gradients into the arena, + bucket
        # all-reduce) runs on the side
        # stream while the main stream continues the dZ chain with dgrad_l -- the two GEMMs are
        # independent given dZ_l, and each hides the other's prologue / epilogue-store bubbles.  Three dZ
        # buffers rotate; a buffer is rewritten only after the wgrad that read it has finished.
        reader_done = {}        # dZ buffer index -> event of the wgrad that reads it
        buf_of_g = None         # index into self.dbuf holding the current dZ (None: self.dy / self.dpre, or planes only)
        nxt = 0
        chain = getattr(self, "chain", set())
        R = self.x.shape[0]
        dzN_ready = dzT_ready = False       # dZ_l already sits in dzw_planes[l] / dzT[l] (written by a split data-gradient epilogue)
        for l in reversed(range(L)):
            lin = self.linears[l]
            inp = self.acts[l - 1] if l > 0 else self.x
            if two:
                side.wait_stream(main)                      # dZ_l is complete on main
                with torch.cuda.stream(side):
                    self._wgrad_layer(l, g, inp, self.wgrad_ws, planes_ready=dzN_ready)
                    if self.buckets is not None:
                        self.buckets.layer_done(L - 1 - l)
                    if buf_of_g is not None:
                        ev = torch.cuda.Event(); ev.record(side); reader_done[buf_of_g] = ev
            else:
                self._wgrad_layer(l, g, inp, self.wgrad_ws, planes_ready=dzN_ready)
                if self.buckets is not None:
                    self.buckets.layer_done(L - 1 - l)
            if l > 0 and l in chain:
                # dZ_{l-1} = (dZ_l W_l) * LeakyReLU'(acts_{l-1}) on the split bodies: T-planes in, gate from the T-planes of the
                # layer input, T-planes out for the next chain layer, N-planes out for the weight gradient of layer l - 1
                if not dzT_ready:
                    ops.mlp_planes_from_f32_t(g, out=self.dzT[l], **self._s16w(1, l))
                prev = (l - 1) in chain
                out = None
                if not prev:
                    if two and nxt in reader_done:
                        main.wait_event(reader_done.pop(nxt))
                    out = self.dbuf[nxt][:, :lin.in_features]
                dxN = self.dzw_planes[l - 1] if self.wide_kinds[l - 1] == 0 else None
                ops.linear_split_dgrad(self.dzT[l], self.wN[l], self.xT[l], self.slope, R, lin.out_features, lin.in_features,
                                       dxT=self.dzT[l - 1] if prev else None, dxN=dxN, dx=out,
                                       **(dict(state=self.s16, layer=l) if self.split_f16_wide else {}))
                dzT_ready, dzN_ready = prev, dxN is not None
                g = out
                if out is not None:
                    buf_of_g = nxt
                    nxt = (nxt + 1) % len(self.dbuf)
                else:
                    buf_of_g = None
            elif l > 0:
                if two and nxt in reader_done:
                    main.wait_event(reader_done.pop(nxt))    # WAR: the wgrad that read this buffer is done
                out = self.dbuf[nxt][:, :lin.in_features]
                ops.linear_dgrad(g, lin.weight, inp, self.slope, out=out)
                g, buf_of_g = out, nxt
                nxt = (nxt + 1) % len(self.dbuf)
                dzN_ready = dzT_ready = False
        if two:
            main.wait_stream(side)
        if self.buckets is not None:
            self.buckets.wait()
        if self.dp and self.s16 is not None and not self._guard_rides:
            # per-layer path: producers run until the last layer's gradient, so the verdict is reduced on its own (4 bytes) behind them
            self.s16.poison_export(self._guard_slot)
            dist.all_reduce(self._guard_slot, op=dist.ReduceOp.SUM, group=self.pg)

    def _adam_folds_into_wgrad(self) -> bool:
        """The optimizer can ride in the weight-gradient reduction when ONE whole-stack split launch produces every gradient of the
        arena and nothing has to happen between gradient and update (no data-parallel all-reduce, no gradient scaling)."""
        if not (self.split_wgrad and self.fused_backward and self.grouped_wgrad) or self.buckets is not None:
            return False
        if self.world * self.dry_ranks != 1:
            return False
        lin_ids = {id(q) for lin in self.linears for q in (lin.weight, lin.bias)}
        return all(lin.bias is not None for lin in self.linears) and all(id(q) in lin_ids for q in self.f.parameters())

    def optimizer_step(self):
        self._packed_current = False
        if getattr(self, "_adam_done", False):       # applied by the weight-gradient reduction of this step (weight_grads)
            self._adam_done = False
            ticked, self._ticked = self._ticked, False
            self._s16_updated = self.s16 is not None
            if not ticked:
                ops.tick(self.step_dev)
            return
        ticked, self._ticked = self._ticked, False
        fused_update = ops.adam_step(self.param_arena, self.grad_arena, self.exp_avg, self.exp_avg_sq, self.step_dev, self.lr,
                                     self.betas[0], self.betas[1], self.eps, grad_scale=1.0 / (self.world * self.dry_ranks),
                                     t_offset=0 if ticked else 1, s16=self.s16)
        self._s16_updated = bool(fused_update)
        if not ticked:
            ops.tick(self.step_dev)

    # -------------------------------------------------------------------------------- whole step
    def _step_body(self, sample: bool):
        # The fragment-order weight copies only depend on the parameters; they are packed IN LINE in front of the forward -- merged with the
        # latent pair draw into one launch where that exists.  (Rounds 2-5 packed on a side stream beside the samplers: the fork / join of
        # a HIP graph costs more than the 9 us of sampling it hid, profiles/r4_DESIGN_history.md.)
        if self.s16 is not None and not self._s16_calibrated:
            self.calibrate_scales(sample)
        self._packed_current = False          # a step always re-packs (parameters may have been set from outside)
        fused = (self.fused_forward or self.fused_backward) and self.device.type == "cuda"
        if not (sample and fused and self._pack_sample_merged()):
            if sample:
                self.sample()
            if fused:
                self.pack()                    # (here, not inside forward(): bench.py's stamps bracket the encoder launch alone)
        st = getattr(self, "stamps", None)         # bench.py: device time stamps around the encoder launches, valid inside the graph
        if st:
            ops.stamp(st["null"], 0); ops.stamp(st["null"], 1)      # empty bracket: the stamp pair's own cost, subtracted by the reader
            ops.stamp(st["mlp_fwd"], 0)
        self.forward()
        if st:
            ops.stamp(st["mlp_fwd"], 1)
        self._in_step = True                   # (loss_forward_backward may leave dy to be finished by this step's backward chain)
        try:
            self.loss_forward_backward()
            self._fold_adam = self._adam_folds_into_wgrad()
            self.backward()
        finally:
            self._fold_adam = self._in_step = False
        self.optimizer_step()
        if self.s16 is not None and not getattr(self, "_s16_updated", False):
            self.s16.update()                  # this step's recorded maxima -> the next step's scales (normally inside the Adam launch)
        self._planes_on_current_scales = False     # (an update has run behind the forward that wrote the planes)

    def calibrate_scales(self, sample: bool = True, passes: Optional[int] = None):
        """f16x2 arithmetic: a launch runs on the scales derived from the PREVIOUS step's maxima, so before the first step (and after
        parameters were replaced from outside) the scales are brought up to the data by un-applied passes: pack, forward, loss and
        backward on the current batch (sampled when `sample`), scale update, no optimizer step; the device step / RNG counter is put
        back, so the first real step draws the very batch it would have drawn.  L + 1 passes: a producer measures its output in fp32
        BEFORE it is cut to fp16, so a pass on scales of 1 gets every activation right, but a gradient of 1e-8 is flushed on its way into
        the next chain link and that link then measures nothing -- each pass settles (at least) one more link of the chain."""
        if self.s16 is None:
            return
        passes = len(self.linears) + 1 if passes is None else int(passes)
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("calibrate_scales() inside a graph capture: call capture(), which calibrates first")
        self._s16_calibrated = True
        tick = self.step_dev.clone()
        for _ in range(passes):
            self._packed_current = False
            if sample:
                self.sample()
            elif self.mix_in_forward:
                self._x_pending = True           # the injected latents go through the mixing prologue again
            self.forward()
            self.loss_forward_backward()
            self.backward()
            self.s16.update()
            self._planes_on_current_scales = False
            self.step_dev.copy_(tick)
            self._ticked = False
        self.s16.clear_flags()                   # the first pass ran on scales of 1: whatever it flagged is not a finding
        self._versions_seen = self._param_versions()
        self._guard_skipped_seen = self.s16.guard()["skipped"]      # (calibration passes the guard withheld are not training steps)

    def _param_versions(self) -> int:
        """Changes when somebody writes the parameters through torch (load_state_dict, copy_, an optimizer of their own): the kernels write
        the arena through raw pointers and leave the version counters alone."""
        return int(self.param_arena._version) + sum(int(q._version) for q in self.f.parameters())

    def _check_external_writes(self):
        """f16x2: parameters replaced from outside since the last look -> the scales are re-measured before the next step.
        Without this the device-side guard would still keep a step on stale scales away from the parameters; this saves its redo steps."""
        if self.s16 is None or not self._watch_versions:
            return
        v = self._param_versions()
        if v != self._versions_seen:
            self._versions_seen = v
            self._s16_calibrated = False
            if self.graph is not None and not torch.cuda.is_current_stream_capturing():
                self.calibrate_scales(True)

    def check_arith(self, raise_on_bug: bool = True) -> dict:
        """LOG-POINT check of the f16x2 arithmetic (host read + sync; a no-op dict for the other arithmetics): the guard's flags, how many
        steps it has withheld so far (`skipped`; `new_skipped` since the last call) and whether the last step is poisoned.  A withheld step
        left parameters, moments and the step / RNG counter untouched and the next step redid its batch on fresh scales, so training has
        lost replays, not correctness; callers that count steps compare with `steps_done`.  Flag bit 2 -- an overflow the update saw but no
        producer announced -- would be a hole in the guard and raises."""
        if self.s16 is None:
            return dict(flags=0, skipped=0, new_skipped=0, poisoned=False, updates=0)
        g = self.s16.guard()
        g["new_skipped"] = g["skipped"] - self._guard_skipped_seen
        self._guard_skipped_seen = g["skipped"]
        if raise_on_bug and (g["flags"] & 4):
            raise _lib.ClicaError("f16x2 arithmetic: the scale update saw an overflow that no producer had announced (flags bit 2): a "
                                  "producer kernel without the guard -- results since the last check are not to be trusted")
        return g

    def arith_state(self) -> dict:
        """Which encoder arithmetic runs and, for f16x2, the state of its scales (host read + sync: log points, tests).  A non-zero
        `flags` means a tensor outgrew its scale by more than 64 x within one step: the step that raised it is not to be trusted."""
        if self.s16 is not None:
            return dict(arith="f16x2" if self.split_f16 else "f16x2 (per-layer wide path; narrow layers native fp32 MFMA)", **self.s16.read())
        if not (self.split_bf16 or getattr(self, "split_wgrad_wide", False)):
            return dict(arith="native_fp32")
        return dict(arith="bf16x3")

    def step(self):
        """One unsupervised step with on-device sampling.  Returns the device tensor
        ``[loss_mean, pos_mean, neg_mean]`` of this rank's rows (no host sync)."""
        self._check_external_writes()
        if self.graph is not None:
            self.graph.replay()
            self._planes_on_current_scales = False
            # a replay updates the parameter arena without passing through ops.adam_step: caches derived from the weights
            # (the drop-in encoder's fragment-order packs, encoders._MLPFusedFn) key on this epoch and must see the change
            ops.PARAM_EPOCH += 1
        else:
            self._step_body(True)
        return self.loss_out[3 * self.B:]

    def step_injected(self, z1, z2):
        self._check_external_writes()
        self.inject(z1, z2)
        self._step_body(False)
        ops.PARAM_EPOCH += 1
        return self.loss_out[3 * self.B:]

    def capture(self, warmup: int = 3):
        """Capture the step into a HIP graph (counters and RNG offsets live on device, so replays
        advance them).  With data parallelism the RCCL collectives (all-gather, reduce-scatter, bucketed
        all-reduce on the side stream) are captured into the same graph; every rank must call capture()
        and then replay in lock-step."""
        # warm-up launches (lazy kernel-attribute setup, allocator) must not count as training: snapshot
        # and restore parameters, optimizer state and the device step / RNG counter around them
        if self.s16 is not None and not self._s16_calibrated:
            self.calibrate_scales(True)
        # (also what makes a step depend on its predecessors besides the parameters: the f16x2 scales, and the loss workspace -- the
        #  matrix-core sweeps build their planes on the grid the previous call measured, csrc/lp_mfma.h)
        state = (self.param_arena, self.exp_avg, self.exp_avg_sq, self.step_dev, self.loss_ws) + ((self.s16.buf,) if self.s16 is not None else ())
        snap = [t.clone() for t in state]
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._step_body(True)
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        for dst, src in zip(state, snap):
            dst.copy_(src)
        self._versions_seen = self._param_versions()      # (the restore wrote the arena through torch: not an external write)
        graph = torch.cuda.CUDAGraph()
        # with collectives in the step, other threads (the process group's watchdog) may legitimately touch the HIP runtime
        # during capture: "thread_local" keeps their calls from invalidating it
        mode = "thread_local" if (self.dp or self.force_collectives) else "global"
        # Explicit begin / end instead of the `torch.cuda.graph` context manager: when the body fails (a collective backend that
        # cannot be captured), that manager raises from capture_end() inside its __exit__ and never restores the stream -- the
        # process is left on a stream that is still capturing and every later CUDA call fails.  Here a failed capture is always
        # terminated (through the HIP runtime if torch refuses), the stream context is always left, and the caller gets the
        # original exception with the device in a usable state (bench.py / train_mlp then run eager launches).
        cap = torch.cuda.Stream(device=self.device)
        cap.wait_stream(torch.cuda.current_stream(self.device))
        try:
            with torch.cuda.stream(cap):
                graph.capture_begin(capture_error_mode=mode)
                try:
                    self._step_body(True)
                    graph.capture_end()
                except BaseException:
                    _abort_capture(graph, cap, (self.side_stream,))
                    if self.side_stream is not None:      # a forked stream cannot be taken out of an invalidated capture: drop it
                        self.side_stream = torch.cuda.Stream(device=self.device)
                    raise
        finally:
            torch.cuda.current_stream(self.device).wait_stream(cap)
        self.graph = graph
        return graph

    def loss_spread(self) -> float:
        """M = log2(e)/tau max_i |y_i - y_0|^2 of the largest embedding cloud the p = 2 matrix-core loss sweeps have seen in this
        trainer (0 on the VALU sweeps): their logit error scales with it (include/clica.h).  Host read + sync -- for
        log points, not for the step."""
        if not self.loss_train:
            return 0.0
        v = C.c_float(0.0)
        _lib.check(_lib.load().clica_lp_loss_train_spread(C.byref(self.desc), self.loss_ws.data_ptr(), self.loss_ws.numel(), C.byref(v),
                                                          _lib.stream_ptr()), "clica_lp_loss_train_spread")
        return float(v.value)

    def loss_guard(self) -> dict:
        """State of the device-side guard of the p = 2 matrix-core loss sweeps (include/clica.h, "THE GUARD"): the largest spread M seen, the
        last step's M, the limit in force and how many steps fell back to the coordinate-difference sweeps.  The decision itself is made
        per step by the kernels (also inside graph replays); this is a host read + sync for log points."""
        if not self.loss_train:
            return dict(max_spread=0.0, last_spread=0.0, limit=0.0, fallback_steps=0)
        v = (C.c_float * 4)()
        _lib.check(_lib.load().clica_lp_loss_train_guard(C.byref(self.desc), self.loss_ws.data_ptr(), self.loss_ws.numel(), v,
                                                         _lib.stream_ptr()), "clica_lp_loss_train_guard")
        return dict(max_spread=float(v[0]), last_spread=float(v[1]), limit=float(v[2]), fallback_steps=int(v[3]))

    def set_loss_matrix_cores(self, on):
        """Switch the p = 2 loss sweeps between the matrix cores for every pool (True) and the coordinate-difference sweeps (False); None = back
        to the default policy (matrix cores only against a pool of >= 4 x the local rows: include/clica.h).  Process-wide (clica_lp_loss_set_matrix_cores); a captured step graph is captured again."""
        _lib.check(_lib.load().clica_lp_loss_set_matrix_cores(-1 if on is None else (2 if on else 0)), "clica_lp_loss_set_matrix_cores")
        if self.graph is not None:
            self.graph = None
            self.capture()

    def plan_summary(self) -> dict:
        """What this rank has planned for its data-parallel step: pool size, workspaces, weight-gradient halves, gradient buckets,
        collectives per step.  bench.py prints it (`ranks` / `dry_ranks`), the dry-run tests assert it."""
        B, n = self.B, self.n
        ranks = self.world * self.dry_ranks
        coll = []
        if self.dp:
            coll.append(dict(op="all_gather", what="embeddings z1_rec", bytes_per_rank=4 * B * n, gathered_bytes=4 * B * n * ranks))
            coll.append(dict(op="all_gather", what="row log-sum-exp", bytes_per_rank=4 * B, gathered_bytes=4 * B * ranks))
            for lo, hi in (self.buckets.buckets if self.buckets is not None else []):
                coll.append(dict(op="all_reduce", what="gradient arena [%d, %d)" % (lo, hi), bytes=4 * (hi - lo)))
        return dict(world=self.world, dry_ranks=self.dry_ranks, planned_ranks=ranks, batch_per_rank=B, pool_rows=int(self.desc.B3),
                    loss_workspace_bytes=int(self.loss_ws.numel()), loss_entry_points="train pair" if self.loss_train else "generic",
                    wgrad_halves=bool(self.wgrad_halves), wgrad_group_workspace_bytes=int(self.group_ws.numel()) if self.group_ws is not None else 0,
                    gradient_buckets=[list(b) for b in (self.buckets.buckets if self.buckets is not None else [])],
                    gradient_arena_elements=int(self.grad_arena.numel()), grad_scale=1.0 / ranks, collectives_per_step=coll,
                    encoder_path="whole-stack" if self.fused_forward else "per-layer", split_bf16=bool(self.split_bf16),
                    graph_captured=self.graph is not None)

    @property
    def steps_done(self) -> int:
        return int(self.step_dev.item())
