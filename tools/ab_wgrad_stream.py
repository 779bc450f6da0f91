#!/usr/bin/env python3
"""A/B of fused_layer.WGRAD_STREAM (the fused layers' weight gradients on a side stream) on the bench step: first that three steps leave
bit-identical parameters in both modes, then interleaved timing (HIP events around 12 steps, 3 rounds each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from afcm_amd import layer_schedule as sched, synthetic
from afcm_amd.networks_stylegan3 import Stylegan3Generator
from afcm_amd.stylegan3_model import StyleGAN3GeneratorStep
from afcm_amd.torch_utils.ops import fused_layer

dev = torch.device('cuda', 0)
inputs = synthetic.generator_inputs(16, size=256, seed=0, device=dev)


def build():
    torch.manual_seed(0)
    G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1, mapping_kwargs=dict(num_layers=8),
                           synthesis_kwargs=dict(sched.DEFAULT_SYNTHESIS_KWARGS, compute_dtype=torch.bfloat16)).to(dev).eval()
    return G, StyleGAN3GeneratorStep(G, lr_G=0.0025, lambda_L1=100.0)


def run(step, n):
    for _ in range(n):
        step.set_input(*inputs)
        step.optimize_parameters()


finals = []
for mode in (False, True):
    fused_layer.WGRAD_STREAM = mode
    G, step = build()
    run(step, 3)
    torch.cuda.synchronize()
    finals.append([p.detach().clone() for p in G.parameters()])
    del G, step
same = all(torch.equal(a, b) for a, b in zip(*finals))
print('# parameters after 3 steps bit-identical in both modes:', same)
del finals
G, step = build()
G.train()
for mode in (False, True):
    fused_layer.WGRAD_STREAM = mode
    run(step, 3)
for rnd in range(3):
    for mode in (False, True):
        fused_layer.WGRAD_STREAM = mode
        run(step, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(step, 12)
        e1.record()
        torch.cuda.synchronize()
        print(f'round {rnd} wgrad on {"a side stream" if mode else "the main stream"}: {e0.elapsed_time(e1) / 12:.3f} ms per step')
