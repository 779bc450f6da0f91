#!/usr/bin/env python3
"""Forward conv on a few hand-picked shapes: tools/conv_shape_probe.py [cin cout size]...  (duration of the 16-bit forward kernel per shape; used with a
rocprofv3 --pmc pass for MFMA-busy share and effective clock)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd.torch_utils.ops import conv2d as C
args = [int(v) for v in sys.argv[1:]] or [512, 512, 62, 512, 512, 126, 512, 512, 84, 362, 512, 148, 128, 128, 254]
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for ci, co, h in zip(args[0::3], args[1::3], args[2::3]):
    x = torch.randn(16, ci, h, h, device='cuda', dtype=torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, device='cuda')
    wp, rp = C.pack_weights(w, torch.bfloat16, 0)
    y = C._conv_raw(x, wp, rp, None, co, 3, 2)
    fl = 2.0 * 16 * co * ci * 9 * y.shape[2] * y.shape[3]
    t = timeit(lambda: C._conv_raw(x, wp, rp, None, co, 3, 2))
    print(f'{ci:3d}->{co:3d} @{h:3d} (out {y.shape[3]})  fwd {t:6.3f} ms {fl/t/1e9:7.1f} TF', flush=True)
