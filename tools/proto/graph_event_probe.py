import torch
x = torch.randn(4096, 4096, device="cuda")
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s):
    y = x @ x
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        e0.record(s)
        y = x @ x
        e1.record(s)
        z = y + 1
        e2.record(s)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
try:
    print("in-graph events:", e0.elapsed_time(e1), e1.elapsed_time(e2))
except Exception as ex:
    print("FAILED:", repr(ex)[:300])
