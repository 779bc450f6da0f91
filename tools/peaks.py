#!/usr/bin/env python3
"""Confirm the chip's practical peaks on the box the bench runs on: HBM copy / read bandwidth with
framework kernels and the dense bf16 / fp16 / fp32 GEMM rate of the vendor library -- context for the roofline fractions in
bench.py, which are quoted against the NOMINAL peaks (8 TB/s, 2.5 PFLOP/s dense 16-bit)."""
import torch


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    p = torch.cuda.get_device_properties(0)
    print(f'{p.name}: {p.multi_processor_count} CUs, {p.total_memory / 2**30:.0f} GiB')
    n = 1 << 30                                    # 1 GiB per operand: far beyond the 256 MB Infinity Cache
    a = torch.empty(n, dtype=torch.uint8, device='cuda').random_(0, 255)
    b = torch.empty_like(a)
    t = timed(lambda: b.copy_(a), 20)
    print(f'HBM copy (1 GiB -> 1 GiB, read + write): {2 * n / t / 1e12:.2f} TB/s')
    af = a.view(torch.float32)
    t = timed(lambda: af.sum(), 20)
    print(f'HBM read (sum over 1 GiB of fp32):       {n / t / 1e12:.2f} TB/s')
    t = timed(lambda: b.fill_(1), 20)
    print(f'HBM write (fill 1 GiB):                  {n / t / 1e12:.2f} TB/s')
    del a, b, af
    for dt, m in ((torch.bfloat16, 8192), (torch.float16, 8192), (torch.float32, 4096)):
        x = torch.randn(m, m, device='cuda', dtype=dt)
        y = torch.randn(m, m, device='cuda', dtype=dt)
        t = timed(lambda: torch.mm(x, y), 10)
        print(f'GEMM {m}^3 {str(dt).split(".")[-1]:9s} (vendor library, random operands): {2 * m ** 3 / t / 1e12:.0f} TFLOP/s')
        x.zero_(); y.zero_()
        t = timed(lambda: torch.mm(x, y), 10)
        print(f'GEMM {m}^3 {str(dt).split(".")[-1]:9s} (vendor library, zero operands):   {2 * m ** 3 / t / 1e12:.0f} TFLOP/s')


if __name__ == '__main__':
    main()
