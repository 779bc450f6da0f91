#!/usr/bin/env python3
"""Run one modulated-conv layer shape fwd+bwd a few times (for rocprofv3)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd.torch_utils.ops.conv2d import modulated_conv2d
ap = argparse.ArgumentParser()
ap.add_argument('--cin', type=int, default=512); ap.add_argument('--cout', type=int, default=512)
ap.add_argument('--hw', type=int, default=36); ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--dtype', default='bf16'); ap.add_argument('--iters', type=int, default=3)
a = ap.parse_args()
dt = {'fp32': torch.float32, 'bf16': torch.bfloat16}[a.dtype]
x = torch.randn(a.batch, a.cin, a.hw, a.hw, device='cuda', dtype=dt).requires_grad_(True)
w = torch.randn(a.cout, a.cin, 3, 3, device='cuda').requires_grad_(True)
s = (torch.randn(a.batch, a.cin, device='cuda') * 0.2 + 1).requires_grad_(True)
for _ in range(a.iters):
    y = modulated_conv2d(x, w, s, padding=2)
    torch.autograd.grad(y.float().sum(), [x, w, s])
torch.cuda.synchronize()
print('done', y.shape)
