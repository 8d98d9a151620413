#!/usr/bin/env python3
"""Is the weight-gradient launch bound by where its operands come from?  The same launch (a) inside the replayed training step, where the
214 MB of plane operands were written by the two encoder launches in front of it, and (b) repeated back to back on the same operands
(they fit the 256 MB Infinity Cache).  HIP events around 20 launches; lr = 0 so that repeating the folded Adam leaves the weights alone.
    python tools/wgrad_warm_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd import encoders
from cl_ica_amd.engine import ContrastiveTrainer, SamplerSpec
n, B = 10, 6144
torch.manual_seed(0)
f = encoders.get_mlp(n, n, [n * 10, n * 50, n * 50, n * 50, n * 50, n * 10]).to("cuda")
gW = torch.randn(3, n, n, device="cuda") / n ** 0.5
tr = ContrastiveTrainer(f, gW, SamplerSpec(n=n), batch_size=B, p=2, lr=0.0, device="cuda")
for _ in range(30):
    tr.step()
torch.cuda.synchronize()
g = tr.y_grad if hasattr(tr, "y_grad") else None
def timed(fn, reps=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    fn(); torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / reps
import inspect
src = inspect.getsource(tr.backward)
gg = tr._head_backward()
tr._fold_adam = tr._adam_folds_into_wgrad()
def wg():
    tr._tail_ready = False
    tr.weight_grads(gg)
print("fold_adam", tr._fold_adam)
print(f"weight gradients + slab reduction, back to back on the same operands: {timed(wg):.1f} us per call")
big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")     # 256 MB: evict the operands between the calls
def wg_cold():
    big.fill_(1)
    wg()
t_fill = timed(lambda: big.fill_(1))
print(f"the same behind a 256 MB fill (operands evicted): {timed(wg_cold) - t_fill:.1f} us per call (fill alone {t_fill:.1f})")
