#!/usr/bin/env python3
"""Memory-format A/B for the two conv configs (MIOpen picks different kernels for NCHW and NHWC): steps/s of bench.py's c4 / c5 steps
with the encoder in contiguous (the reference's layout) and channels_last format.   python tools/conv_layout_probe.py   (GPU box)"""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl_ica_amd.optim import Adam
dev = torch.device("cuda")


def timeit(step, steps=15, warm=6):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record(); torch.cuda.synchronize()
    return steps / (e0.elapsed_time(e1) * 1e-3)


for fmt_name, fmt in (("contiguous (NCHW)", torch.contiguous_format), ("channels_last (NHWC)", torch.channels_last)):
    from cl_ica_amd import threedident as T
    torch.manual_seed(0)
    a = types.SimpleNamespace(position_only=True, rotation_and_color_only=False, rotation_only=False, color_only=False,
                              non_periodic_rotation_and_color=False, box_constraint="fix", sphere_constraint=None,
                              unsupervised_loss="l2", identity_solution=False, encoder="rn18")
    f = T.setup_f(a, 3, 0).to(dev).to(memory_format=fmt); f.train()
    loss = T.make_unsupervised_loss(a, 3); opt = Adam(f.parameters(), lr=1e-4)
    x1 = torch.randn(1024, 3, 64, 64, device=dev).contiguous(memory_format=fmt); x2 = (x1 + 0.1 * torch.randn_like(x1)).contiguous(memory_format=fmt)
    r = timeit(lambda: T.train_step(((None, None), (x1, x2)), loss, opt, f, sync=False))
    print(f"c4 ResNet-18 {fmt_name}: {r:.1f} steps/s", flush=True)
    del f, opt; torch.cuda.empty_cache()
    from cl_ica_amd.kitti_masks.solver import Solver
    sa = types.SimpleNamespace(cuda=True, ckpt_dir="/tmp", output_dir="/tmp", dataset="kitti", max_iter=1, z_dim=5, num_channel=1, lr=1e-4,
                               beta1=0.9, beta2=0.999, ckpt_name="last", log_step=1000, save_step=10 ** 9, box_norm=True, p=1)
    S = Solver(sa, None); S.net_mode(train=True)
    S.net.to(memory_format=fmt)
    x = (torch.rand(2048, 1, 64, 64, device=dev) < 0.1).float().contiguous(memory_format=fmt)
    r = timeit(lambda: S.train_iteration(x), steps=40)
    print(f"c5 BetaVAE_H {fmt_name}: {r:.1f} steps/s", flush=True)
    del S; torch.cuda.empty_cache()
