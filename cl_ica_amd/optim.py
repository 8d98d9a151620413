"""Adam over ONE flat parameter arena, one HIP launch per step (csrc/adam.hip: ``clica_adam_step``).

Counterpart of the ``torch.optim.Adam(f.parameters(), lr=...)`` the reference's drivers build (main_mlp.py:312,
main_3dident.py:447-448, kitti_masks/solver.py:36-40) with the same update rule (bias-corrected, eps outside the square
root, no weight decay / amsgrad) and the same ``state_dict()`` layout (``state[i] = {step, exp_avg, exp_avg_sq}``,
``param_groups``), so optimizer checkpoints interchange with the reference's.

Construction re-points every parameter's ``.data`` into a 16-byte aligned slice of one contiguous fp32 arena and installs
``.grad`` views into a matching gradient arena (autograd accumulates into them in place), so ``zero_grad()`` is one memset,
``step()`` one kernel launch, and a data-parallel run all-reduces ONE buffer (``all_reduce_grads``).
"""
from __future__ import annotations

import weakref
from typing import Iterable, Optional

import torch
import torch.distributed as dist

from . import lazy, ops
from ._lib import require_cuda

__all__ = ["Adam"]


class Adam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group: Optional[dist.ProcessGroup] = None):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        for p in self.params:
            require_cuda(p.data, "parameter")
        dev = self.params[0].device
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.param_groups = [dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False,
                                  params=list(range(len(self.params))))]
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.offsets, self.total = offs, total
        self.param_arena = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad_arena = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._ticket = torch.zeros(1, dtype=torch.int32, device=dev)       # clica_adam_step_tick: the update's last workgroup advances step_dev
        for p, off in zip(self.params, offs):
            view = self.param_arena[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad_arena[off:off + p.numel()].view(p.shape)
            # the drop-in encoder's backward (encoders._MLPFusedFn) adds its dW / db straight into this view (and hands autograd
            # None) when it finds it installed: no per-parameter gradient tensors, no AccumulateGrad launches
            p._clica_grad_view = p.grad
            p._clica_flat_opt = weakref.ref(self)      # (encoders._s16_ctx: the f16x2 arithmetic needs THIS optimizer to apply the step)
        self._s16 = None
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1

    # ------------------------------------------------------------------ torch.optim surface
    def zero_grad(self, set_to_none: bool = False):
        """One memset of the gradient arena (the ``.grad`` views stay installed; ``set_to_none`` is ignored on purpose)."""
        self.grad_arena.zero_()
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad_arena.data_ptr() + 4 * off:
                p.grad = self.grad_arena[off:off + p.numel()].view(p.shape)      # someone set it to None / replaced it
                p._clica_grad_view = p.grad

    def _bind_s16(self, ctx) -> bool:
        """An encoder's f16x2 context asks to ride in this optimizer's launch (scale update + guard).  One at a time: the launch takes one."""
        cur = self._s16() if self._s16 is not None else None
        if cur is None or cur is ctx:
            self._s16 = weakref.ref(ctx)
            return True
        return False

    def all_reduce_grads(self):
        """Data parallel: sum the gradient arena over the ranks (the 1/world average is applied inside ``step``)."""
        self._adopt_foreign_grads()      # `model.zero_grad()` (set_to_none) makes autograd allocate .grad OUTSIDE the arena: bring
        if self.world > 1:               # those back BEFORE the exchange, or the stale arena is what gets summed
            dist.all_reduce(self.grad_arena, op=dist.ReduceOp.SUM, group=self.pg)

    def _adopt_foreign_grads(self):
        """``nn.Module.zero_grad()`` / ``zero_grad(set_to_none=True)`` drop the arena views, after which autograd allocates
        fresh ``.grad`` tensors outside the arena.  Bring those back (copy + re-install the view) so the launch below
        reads the gradients autograd produced; a parameter that received no gradient this step contributes zeros."""
        base = self.grad_arena.data_ptr()
        for p, off in zip(self.params, self.offsets):
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * off:
                continue                                   # the installed view (the common case: no tensor is built for it)
            view = self.grad_arena[off:off + p.numel()].view(p.shape)
            if g is None:
                view.zero_()
            else:
                view.copy_(g)
            p.grad = view
            p._clica_grad_view = view

    def step(self, closure=None):
        """One launch over the whole arena.  Differences from ``torch.optim.Adam`` (none of the reference's drivers can
        observe them): ONE parameter group; one global step count; a parameter without a gradient is updated with g = 0
        (its moments decay) where torch would skip it."""
        loss = closure() if closure is not None else None
        lazy.flush_all()          # a deferred encoder call (cl_ica_amd/lazy.py) is computed with the parameters it was made with
        if len(self.param_groups) != 1:
            raise ValueError("the flat Adam supports exactly one parameter group")
        self._adopt_foreign_grads()
        g = self.param_groups[0]
        s16 = self._s16() if self._s16 is not None else None
        if s16 is not None:
            # the encoder runs in the f16x2 arithmetic: its scale update rides in this launch, and a step whose producers met a value
            # beyond its scale (the guard, include/clica.h) leaves parameters and moments untouched and takes the step count back
            # (one launch: update, scale update, guard, and -- its last workgroup -- the counter's tick)
            ops.adam_step(self.param_arena, self.grad_arena, self.exp_avg, self.exp_avg_sq, self.step_dev, float(g["lr"]),
                          float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), grad_scale=1.0, t_offset=1, s16=s16.state,
                          ticket=self._ticket)
        else:
            ops.adam_step(self.param_arena, self.grad_arena, self.exp_avg, self.exp_avg_sq, self.step_dev, float(g["lr"]),
                          float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), grad_scale=1.0 / self.world, ticket=self._ticket)
        lazy.after_step()         # e.g. the encoder's fragment-order weight copies, re-packed now rather than in front of the next forward
        return loss

    # ------------------------------------------------------------------ checkpoints (torch.optim.Adam layout)
    def state_dict(self):
        step = int(self.step_dev.item())
        state = {}
        if step > 0:
            for i, (p, off) in enumerate(zip(self.params, self.offsets)):
                sl = slice(off, off + p.numel())
                state[i] = dict(step=torch.tensor(float(step)), exp_avg=self.exp_avg[sl].view(p.shape).clone(),
                                exp_avg_sq=self.exp_avg_sq[sl].view(p.shape).clone())
        return dict(state=state, param_groups=[dict(self.param_groups[0])])

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.param_groups[0].update(lr=float(g["lr"]), betas=tuple(g["betas"]), eps=float(g["eps"]))
        steps = set()
        for i, st in sd["state"].items():
            p, off = self.params[int(i)], self.offsets[int(i)]
            sl = slice(off, off + p.numel())
            self.exp_avg[sl].view(p.shape).copy_(st["exp_avg"])
            self.exp_avg_sq[sl].view(p.shape).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; the flat Adam keeps one step counter")
        self.step_dev.fill_(steps.pop() if steps else 0)
