#!/usr/bin/env python3
"""Forward-only timing of one filtered_lrelu layer with and without the sign tensor (experiment aid)."""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import layer_schedule as sched
from afcm_amd.torch_utils.ops import filtered_lrelu as flr
ap = argparse.ArgumentParser(); ap.add_argument('--layer', default='encoder_1'); ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()
pl = sched.plan(256, 4, 1, {})
L = [l for l in pl['enc'] + pl['dec'] if l['name'] == a.layer][0]
h = L['in_size'] + 2
kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=math.sqrt(2), slope=0.2, clamp=256.0)
fu, fd = L['fu'].cuda(), L['fd'].cuda()
for grad in (True, False):
    x = torch.randn(16, L['cout'], h, h, device='cuda', dtype=torch.bfloat16).requires_grad_(grad)
    with torch.set_grad_enabled(grad):
        for _ in range(3): y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=None, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=None, **kw)
        e1.record(); torch.cuda.synchronize()
    print(f'{a.layer} signs={"WRITE" if grad else "NONE"}: {e0.elapsed_time(e1) / a.iters:.3f} ms')
