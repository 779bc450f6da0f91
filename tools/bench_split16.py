"""Timing of the fp32 -> split-operand passes on one activation tensor: amax_bits, split16, torch.aminmax (GPU box, repo root;
PYTHONPATH=. python tools/bench_split16.py)."""
import torch
from afcm_amd.torch_utils.ops import conv2d as C


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for shape in ((16, 64, 278, 278), (16, 128, 278, 278), (16, 512, 86, 86), (16, 512, 38, 38), (512, 512, 3, 3)):
    x = torch.randn(shape, device='cuda')
    mb = x.numel() * 4 / 1e6
    t_amax = timed(lambda: C.amax_bits(x))
    f = C.amax_bits(x)
    t_split = timed(lambda: C.split16(x, None, 2, torch.float16, f)) if x.shape[3] % 2 == 0 else float('nan')
    t_torch = timed(lambda: torch.aminmax(x))
    print(f'{str(shape):22s} {mb:8.1f} MB  amax_bits {t_amax:7.1f} us ({mb / t_amax:6.2f} TB/s)  split16 {t_split:7.1f} us ({mb * 2 / t_split:6.2f} TB/s r+w)  '
          f'torch.aminmax {t_torch:7.1f} us ({mb / t_torch:6.2f} TB/s)')
