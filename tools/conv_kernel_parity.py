#!/usr/bin/env python3
"""Parity of the 16-bit 3x3 conv kernels on a few shapes against an fp32 torch reference of the same op (GPU): the kernel is chosen
by the environment (AFCM_CONV_GATHER=1 / AFCM_CONV_DIRECT=1 / nothing), read once per process -- tests/test_gpu_conv.py runs this
script in a child process per experimental kernel.  Exit code 0 = every case within tolerance."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd.torch_utils.ops import conv2d as C

def main():
    torch.manual_seed(0)
    worst = 0.0
    # (n, cin, cout, h, w, pad, dtype): 128-row blocks (cout > 64, multiple of 128 after padding), channel tails (cin % 16 != 0), edge tiles,
    # both paddings the generator uses (2: forward, 0: data gradient), a plane narrower than one piece column
    cases = [(2, 32, 128, 38, 38, 2), (2, 37, 128, 54, 54, 2), (1, 64, 256, 150, 150, 2), (2, 91, 128, 86, 86, 0), (1, 128, 128, 278, 278, 2),
             (2, 16, 128, 10, 6, 2), (1, 48, 384, 62, 62, 0)]
    for dtype, tol in ((torch.bfloat16, 2e-2), (torch.float16, 3e-3)):
        for n, ci, co, h, w, pad in cases:
            x = torch.randn(n, ci, h, w, device='cuda').to(dtype)
            wt = torch.randn(co, ci, 3, 3, device='cuda') / (3 * ci ** 0.5)
            wp, rp = C.pack_weights(wt, dtype, 0)
            y = C._conv_raw(x, wp, rp, None, co, 3, pad).float()
            ref = F.conv2d(x.float(), wt.to(dtype).float(), padding=pad)
            err = (y - ref).abs().max().item() / max(1e-6, ref.abs().max().item())
            worst = max(worst, err)
            ok = err <= tol and torch.isfinite(y).all().item()
            print(f'{str(dtype):15s} n{n} {ci:3d}->{co:3d} {h}x{w} pad {pad}: rel max err {err:.2e} {"ok" if ok else "FAIL"}', flush=True)
            if not ok:
                return 1
    print('worst', worst)
    return 0

if __name__ == '__main__':
    sys.exit(main())
