#!/usr/bin/env python3
"""Time the per-plane dot product kernel (C ABI afcm_plane_dot) on the generator's activation shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from afcm_amd.torch_utils.ops.conv2d import plane_dot

for c, hw in ((64, 278), (64, 276), (181, 148), (512, 86), (512, 84), (512, 52), (512, 36)):
    a = torch.randn(16, c, hw, hw, device='cuda').to(torch.bfloat16)
    b = torch.randn(16, c, hw, hw, device='cuda').to(torch.bfloat16)
    big = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    for _ in range(3):
        plane_dot(a, b)
    ts = []
    for _ in range(10):
        big.zero_()                      # evict a / b from the 256 MB infinity cache
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plane_dot(a, b); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    print(f'C={c:4d} {hw:3d}^2  {ms * 1e3:7.1f} us  {2 * a.numel() * 2 / ms / 1e6:7.1f} GB/s')
