#!/usr/bin/env python3
"""Count the framework (aten) operators one generator training step launches -- the launch-bound glue around the HIP kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from afcm_amd import layer_schedule as sched, synthetic
from afcm_amd.networks_stylegan3 import Stylegan3Generator
from afcm_amd.stylegan3_model import StyleGAN3GeneratorStep

torch.manual_seed(0)
G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1, mapping_kwargs=dict(num_layers=8),
                       synthesis_kwargs=dict(sched.DEFAULT_SYNTHESIS_KWARGS, compute_dtype=torch.bfloat16)).cuda().train()
step = StyleGAN3GeneratorStep(G)
a, b, z, c = synthetic.generator_inputs(16, size=256, seed=0, device='cuda')
for _ in range(3):
    step.set_input(a, b, z, c); step.optimize_parameters()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step.set_input(a, b, z, c); step.optimize_parameters()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0]
rows.sort(key=lambda r: -r[2])
for k, n, t in rows[:45]:
    print(f'{k[:70]:70s} {n:5d} {t / 1e3:8.3f} ms')
