#!/usr/bin/env python3
"""Pad-1 3x3 weight gradient on the discriminator's shapes: the dword LDS-DMA kernel against the pad-2 granule kernel on a zero-framed dy
(conv2d._frame_dy; the frame is one extra pass over dy)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd.torch_utils.ops import conv2d as C
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for res, c in ((256, 64), (128, 128), (64, 256), (32, 512)):
    x = torch.randn(16, c, res, res, device='cuda', dtype=torch.bfloat16)
    dy = torch.randn(16, c, res, res, device='cuda', dtype=torch.bfloat16)
    out = []
    for lim in (0, 1 << 30):
        C._FRAME_WGRAD_MAX = lim
        out.append(timeit(lambda: C._wgrad_raw(dy, x, c, c, 3, 1)))
    fl = 2.0 * 16 * c * c * 9 * res * res
    print(f'{res:4d}^2 x {c:3d}: dword kernel {out[0]*1e3:7.1f} us ({fl/out[0]/1e9:6.0f} TF)   framed + granule kernel {out[1]*1e3:7.1f} us ({fl/out[1]/1e9:6.0f} TF)')
