#!/usr/bin/env python3
"""Time the weight pack (C ABI afcm_conv2d_pack_weights2: forward + data-gradient image in one launch) on the generator's layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from afcm_amd.torch_utils.ops.conv2d import pack_weights_both

for o, i in ((64, 4), (64, 64), (128, 91), (181, 128), (512, 362), (512, 512)):
    w = torch.randn(o, i, 3, 3, device='cuda')
    for _ in range(3):
        pack_weights_both(w, torch.bfloat16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        pack_weights_both(w, torch.bfloat16)
    e1.record(); torch.cuda.synchronize()
    print(f'{i:4d}->{o:4d}  {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per launch (both images)')
