#!/usr/bin/env python3
"""Does the output row alignment bound the 16-bit filtered_lrelu kernels?  enc0's configuration (up 2 / down 2, 64 channels, batch 16) on
square planes whose output width is 256 / 276 / 312 / 320: 256 and 320 give 128-byte-aligned output rows and whole 64-column groups,
312 has nearly whole groups (97.5 %) but misaligned rows, 276 is the generator's own (86 % of 5 groups, misaligned)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import layer_schedule as sched  # noqa: E402
from afcm_amd.torch_utils.ops import filtered_lrelu as flr  # noqa: E402

pl = sched.plan(256, 4, 1, {})
L = pl['enc'][int(os.environ.get('LAYER', '1'))]
dt = torch.bfloat16
for out in [int(v) for v in (sys.argv[1:] or ['256', '276', '312', '320'])]:
    h = out + (L['in_size'] + L['k'] - 1 - L['out_size'])
    x = torch.randn(16, 64, h, h, device='cuda', dtype=dt).requires_grad_(True)
    kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=math.sqrt(2), slope=0.2, clamp=256.0)
    fu, fd = L['fu'].cuda(), L['fd'].cuda()
    y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=None, **kw)
    signs = y.grad_fn.saved_tensors[2]
    r = torch.randn_like(y)
    nbytes = (x.numel() + y.numel()) * 2 + signs.numel()
    tf = tb = 1e9
    for rep in range(3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for _ in range(2):
            y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=None, **kw)
        ev[0].record()
        for _ in range(20):
            y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=None, **kw)
        ev[1].record()
        for _ in range(2):
            torch.autograd.grad(y, x, r, retain_graph=True)
        ev[2].record()
        for _ in range(20):
            torch.autograd.grad(y, x, r, retain_graph=True)
        ev[3].record()
        torch.cuda.synchronize()
        tf = min(tf, ev[0].elapsed_time(ev[1]) / 20)
        tb = min(tb, ev[2].elapsed_time(ev[3]) / 20)
    print(f'in {h:3d} -> out {y.shape[-1]:3d}  ({nbytes/1e6:6.1f} MB)  fwd {tf:.3f} ms {nbytes/tf/1e6:7.1f} GB/s   bwd {tb:.3f} ms {nbytes/tb/1e6:7.1f} GB/s', flush=True)
