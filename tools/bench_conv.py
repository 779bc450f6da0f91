#!/usr/bin/env python3
"""Per-layer conv micro-benchmark: forward, data gradient, weight gradient TFLOP/s for every conv of the 256^2 generator."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import layer_schedule as sched
from afcm_amd.torch_utils.ops import conv2d as C
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16); ap.add_argument('--dtype', default='bf16'); ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--zeros', action='store_true', help='all-zero operands: same cycles, higher clock where the chip is power-managed')
ap.add_argument('--pitched', action='store_true', help='row-pitched operands and results, as the fused layer node drives the kernels (afcm_amd/torch_utils/ops/_rows.py)')
a = ap.parse_args()
dt = {'fp32': torch.float32, 'bf16': torch.bfloat16}[a.dtype]
pl = sched.plan(256, 4, 1, {})
seen = set(); tot = [0, 0, 0, 0.0]
def timeit(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters
for L in pl['enc'] + pl['dec']:
    key = (L['cin'], L['cout'], L['in_size'], L['k'])
    n, ci, co, h, k = a.batch, L['cin'], L['cout'], L['in_size'], L['k']
    pad = k - 1
    x = torch.randn(n, ci, h, h, device='cuda', dtype=dt)
    w = torch.randn(co, ci, k, k, device='cuda')
    if a.zeros:
        x.zero_(); w.zero_()
    wp, rp = C.pack_weights(w, dt, 0); wpt, rpt = C.pack_weights(w, dt, 1)
    if a.pitched:
        from afcm_amd.torch_utils.ops import _rows
        xp = _rows.empty(list(x.shape), x.dtype, x.device); xp.copy_(x); x = xp
    y = C._conv_raw(x, wp, rp, None, co, k, pad, pitched_out=a.pitched)
    fl = 2.0 * n * co * ci * k * k * y.shape[2] * y.shape[3]
    tf = timeit(lambda: C._conv_raw(x, wp, rp, None, co, k, pad, pitched_out=a.pitched))
    td = timeit(lambda: C._conv_raw(y, wpt, rpt, None, ci, k, k - 1 - pad, pitched_out=a.pitched))
    tw = timeit(lambda: C._wgrad_raw(y, x, co, ci, k, pad))
    tot[0] += tf; tot[1] += td; tot[2] += tw; tot[3] += fl
    if key not in seen:
        seen.add(key)
        print(f'{L["name"]:14s} {ci:3d}->{co:3d} @{h:3d}  fwd {tf:6.3f} ms {fl/tf/1e9:7.1f} TF  dgrad {td:6.3f} ms {fl/td/1e9:7.1f} TF  wgrad {tw:6.3f} ms {fl/tw/1e9:7.1f} TF')
print(f'TOTAL fwd {tot[0]:.1f} ms {tot[3]/tot[0]/1e9:.0f} TF | dgrad {tot[1]:.1f} ms {tot[3]/tot[1]/1e9:.0f} TF | wgrad {tot[2]:.1f} ms {tot[3]/tot[2]/1e9:.0f} TF')
