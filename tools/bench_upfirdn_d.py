#!/usr/bin/env python3
"""upfirdn2d on the discriminator's shapes (batch 16, bf16 blocks): blur ahead of the stride-2 conv (pad 2,2 -> res + 1) and the skip branch's
blur + down 2 -- forward and the gradient pass, achieved GB/s of the algorithmic bytes (read x + write y)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd.torch_utils.ops import upfirdn2d
f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for dt in (torch.bfloat16, torch.float32):
    for res, c in ((256, 64), (128, 128), (64, 256), (32, 512), (16, 512)):
        x = torch.randn(16, c, res, res, device='cuda', dtype=dt).requires_grad_(True)
        for name, kw in (('blur pad 2,3', dict(padding=[2, 3, 2, 3])), ('blur down 2', dict(down=2, padding=[1, 1, 1, 1]))):
            y = upfirdn2d.upfirdn2d(x, f, **kw)
            g = torch.randn_like(y)
            tf = timeit(lambda: upfirdn2d.upfirdn2d(x.detach(), f, **kw))
            tb = timeit(lambda: torch.autograd.grad(upfirdn2d.upfirdn2d(x, f, **kw), x, g)) - tf
            mb = (x.numel() + y.numel()) * x.element_size() / 1e6
            print(f'{str(dt)[6:]:9s} {res:4d}^2 x {c:3d}  {name:12s} out {tuple(y.shape[2:])}  fwd {tf*1e3:7.1f} us {mb/tf:7.0f} GB/s   bwd {tb*1e3:7.1f} us {mb/tb:7.0f} GB/s')
