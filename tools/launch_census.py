#!/usr/bin/env python3
"""Which Python lines launch the step's kernels: one generator training step (bench.py's default workload) under torch.profiler with
stacks; prints the launches per step by kernel name, and for the FRAMEWORK kernels (fills, copies, elementwise, reductions, GEMMs --
everything that is not an afcm_* kernel) the aten op and the innermost afcm_amd source line that asked for it.
    python tools/launch_census.py [--dtype bf16] [--batch 16] [--top 60]"""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import layer_schedule as sched, synthetic
from afcm_amd.networks_stylegan3 import Stylegan3Generator
from afcm_amd.stylegan3_model import StyleGAN3GeneratorStep

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16'); ap.add_argument('--batch', type=int, default=16); ap.add_argument('--top', type=int, default=80)
a = ap.parse_args()
dt = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[a.dtype]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1, mapping_kwargs=dict(num_layers=8),
                       synthesis_kwargs=dict(sched.DEFAULT_SYNTHESIS_KWARGS, compute_dtype=dt)).to(dev).train()
step = StyleGAN3GeneratorStep(G, lr_G=0.0025, lambda_L1=100.0)
real_A, real_B, z, c = synthetic.generator_inputs(a.batch, size=256, seed=0, device=dev)


def one():
    step.set_input(real_A, real_B, z, c)
    step.optimize_parameters()


for _ in range(3):
    one()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    one()
    torch.cuda.synchronize()
ev = prof.events()
kernels = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA and 'Memcpy' not in e.name and 'Memset' not in e.name]
memops = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA and ('Memcpy' in e.name or 'Memset' in e.name)]
print(f'# one step: {len(kernels)} kernel launches, {len(memops)} memcpy / memset operations')
by = collections.Counter(e.name.split('(')[0][:90] for e in kernels)
afcm = sum(n for k, n in by.items() if 'afcm' in k)
print(f'# afcm kernels {afcm}, framework kernels {len(kernels) - afcm}')
for k, n in by.most_common(a.top):
    print(f'{n:5d}  {k}')
# CPU-side ops that launched framework kernels: aten op -> innermost afcm_amd frame
cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
sites = collections.Counter()
for e in cpu:
    if not e.name.startswith('aten::') or not getattr(e, 'kernels', None):
        continue
    nk = len([k for k in e.kernels if 'afcm' not in k.name])
    if nk == 0:
        continue
    # only leaf aten ops (a parent op lists its children's kernels as well)
    if any(ch.name.startswith('aten::') and getattr(ch, 'kernels', None) for ch in (e.cpu_children or [])):
        continue
    frame = next((f for f in (e.stack or []) if 'afcm_amd' in f or 'bench.py' in f), (e.stack or ['?'])[0] if e.stack else '?')
    sites[(e.name, frame.strip()[-110:])] += nk
print('# framework launches by (aten op, innermost afcm_amd frame)')
for (op, fr), n in sites.most_common(a.top):
    print(f'{n:5d}  {op:28s} {fr}')
