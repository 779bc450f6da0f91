#!/usr/bin/env python3
"""Probe (GPU box): N gloo ranks on ONE device, K async all-reduces of GPU tensors in flight, then wait -- separates a limitation of the rehearsal
transport (gloo staging CUDA tensors of several processes that share a device) from the bucket logic of afcm_amd/distributed.py.
torchrun --nproc-per-node N tools/gloo_gpu_probe.py [K] [MB]"""
import faulthandler, os, sys, time
import torch, torch.distributed as dist
faulthandler.dump_traceback_later(60, exit=True)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 13
mb = float(sys.argv[2]) if len(sys.argv) > 2 else 25
dist.init_process_group('gloo')
r, w = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
bufs = [torch.full([int(mb * 1e6 / 4)], float(r + 1), device='cuda') for _ in range(k)]
torch.cuda.synchronize()
for it in range(3):
    t0 = time.time()
    hs = [dist.all_reduce(b, async_op=True) for b in bufs]
    for h in hs:
        h.wait()
    torch.cuda.synchronize()
    if r == 0:
        print(f'iteration {it}: {k} async all-reduces of {mb} MB over {w} gloo ranks on one device: {time.time() - t0:.2f} s, value {bufs[0][0].item():.0f}', flush=True)
dist.destroy_process_group()
