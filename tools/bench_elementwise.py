#!/usr/bin/env python3
"""Bandwidth of the small HBM-bound kernels on the shapes the steps run them at (batch 16): bias_act (the discriminator's lrelu + bias + clamp),
scale_planes, plane_dot -- achieved GB/s of the algorithmic bytes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd.torch_utils.ops import bias_act, conv2d as C
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for dt in (torch.bfloat16, torch.float32):
    for res, c in ((256, 64), (128, 128), (64, 256), (32, 512), (276, 128), (148, 362)):
        x = torch.randn(16, c, res, res, device='cuda', dtype=dt)
        b = torch.randn(c, device='cuda', dtype=dt)
        s = torch.rand(16, c, device='cuda')
        mb = x.numel() * x.element_size() / 1e6
        t = timeit(lambda: bias_act.bias_act(x, b, act='lrelu', gain=1.4, clamp=256.0))
        print(f'{str(dt)[6:]:9s} {res:4d}^2 x {c:3d}  bias_act lrelu fwd   {t*1e3:7.1f} us {2*mb/t:7.0f} GB/s')
        xg = x.clone().requires_grad_(True)
        y = bias_act.bias_act(xg, b, act='lrelu', gain=1.4, clamp=256.0)
        g = torch.randn_like(y)
        tb = timeit(lambda: torch.autograd.grad(y, xg, g, retain_graph=True))
        print(f'{str(dt)[6:]:9s} {res:4d}^2 x {c:3d}  bias_act lrelu bwd   {tb*1e3:7.1f} us {3*mb/tb:7.0f} GB/s  (reads dy + y, writes dx)')
        if dt != torch.float32:
            t = timeit(lambda: C.scale_planes(x, s))
            print(f'{str(dt)[6:]:9s} {res:4d}^2 x {c:3d}  scale_planes         {t*1e3:7.1f} us {2*mb/t:7.0f} GB/s')
            t = timeit(lambda: C.plane_dot(x, x))
            print(f'{str(dt)[6:]:9s} {res:4d}^2 x {c:3d}  plane_dot(a, b)      {t*1e3:7.1f} us {2*mb/t:7.0f} GB/s')
