#!/usr/bin/env python3
"""plane_dot bandwidth probe: (a) does the offset between the two operands matter (HBM channel aliasing)? -- no; (b) cold
(after a 512 MB fill that leaves dirty lines in front of it) vs back-to-back launches: 2.7 vs 6.2 TB/s -- the isolated figure
measures the eviction fill's write-back, not the kernel."""
import os, sys
sys.path.insert(0, '/root/repo')
import torch
from afcm_amd.torch_utils.ops.conv2d import plane_dot
c, hw = 64, 278
n = 16 * c * hw * hw
a = torch.randn(16, c, hw, hw, device='cuda').to(torch.bfloat16)
buf = torch.randn(n + (1 << 22), device='cuda').to(torch.bfloat16)
big = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
print('a ptr %x' % a.data_ptr())
for off in (0, 128, 1024, 2048 + 128, 4096 + 256, 65536 + 1024, (1 << 20) + 4096 + 512):
    b = buf[off:off + n].view(16, c, hw, hw)
    for _ in range(3): plane_dot(a, b)
    ts = []
    for _ in range(10):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plane_dot(a, b); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[5]
    print(f'off {off:8d} elems  b ptr {b.data_ptr():x}  {ms*1e3:7.1f} us {2*n*2/ms/1e6:7.1f} GB/s')
# single stream: plane sums
ts = []
for _ in range(10):
    big.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plane_dot(a); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[5]
print(f'single stream {ms*1e3:7.1f} us {n*2/ms/1e6:7.1f} GB/s')
# without eviction (infinity cache warm? 316 MB > 256 MB)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): plane_dot(a, b)
e1.record(); torch.cuda.synchronize()
print(f'back-to-back {e0.elapsed_time(e1)/20*1e3:7.1f} us')
