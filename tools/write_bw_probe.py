"""How fast does this MI355X take plain writes?  (context for the first conv stage's 268 MB output: DESIGN 4.3e)"""
import torch
dev = torch.device("cuda")
def timeit(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for mb in (67, 268, 1072):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    t_fill = timeit(lambda: x.fill_(1.0))
    t_copy = timeit(lambda: y.copy_(x))
    t_read = timeit(lambda: x.sum())
    print(f"{mb} MB: fill {t_fill:.1f} us = {mb * 1.048576 / t_fill:.2f} TB/s written | copy {t_copy:.1f} us = {2 * mb * 1.048576 / t_copy:.2f} TB/s moved | sum {t_read:.1f} us = {mb * 1.048576 / t_read:.2f} TB/s read")
    del x, y
