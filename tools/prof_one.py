#!/usr/bin/env python3
"""Run one filtered_lrelu layer shape a few times (for rocprofv3 kernel-trace / PMC passes)."""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import layer_schedule as sched
from afcm_amd.torch_utils.ops import filtered_lrelu as flr

ap = argparse.ArgumentParser()
ap.add_argument('--layer', default='encoder_1')
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--dtype', default='fp32')
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--bwd', type=int, default=1)
ap.add_argument('--no-bias', action='store_true')
a = ap.parse_args()
dt = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}[a.dtype]
pl = sched.plan(256, 4, 1, {})
L = [l for l in pl['enc'] + pl['dec'] if l['name'] == a.layer][0]
h = L['in_size'] + L['k'] - 1
x = torch.randn(a.batch, L['cout'], h, h, device='cuda', dtype=dt).requires_grad_(True)
b = None if a.no_bias else torch.zeros(L['cout'], device='cuda', dtype=dt)
kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=math.sqrt(2), slope=0.2, clamp=256.0)
fu, fd = L['fu'].cuda(), L['fd'].cuda()
for _ in range(a.iters):
    y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=b, **kw)
    if a.bwd:
        torch.autograd.grad(y, x, torch.ones_like(y))
torch.cuda.synchronize()
print('done', y.shape)
