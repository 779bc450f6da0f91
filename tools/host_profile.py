#!/usr/bin/env python3
"""Where the HOST time of a generator training step goes: cProfile over 6 steps of bench.py's default workload, once the GPU queue is
kept short (a synchronize per step, so that the profile shows the launch path, not a full queue).  Prints the functions by own time and
by cumulative time."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '.')
from afcm_amd import layer_schedule as sched, synthetic
from afcm_amd.networks_stylegan3 import Stylegan3Generator
from afcm_amd.stylegan3_model import StyleGAN3GeneratorStep
torch.set_num_threads(4)
dev = torch.device('cuda', 0)
kw = dict(sched.DEFAULT_SYNTHESIS_KWARGS)
G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1,
                       mapping_kwargs=dict(num_layers=8), synthesis_kwargs=dict(kw, compute_dtype=torch.bfloat16)).to(dev).train()
step = StyleGAN3GeneratorStep(G, lr_G=0.0025, lambda_L1=100.0)
a, b, z, c = synthetic.generator_inputs(16, size=256, seed=0, device=dev)
for _ in range(3):
    step.set_input(a, b, z, c); step.optimize_parameters()
torch.cuda.synchronize()
t0 = time.perf_counter(); host = 0.0
for _ in range(6):
    h0 = time.perf_counter()
    step.set_input(a, b, z, c); step.optimize_parameters()
    host += time.perf_counter() - h0
    torch.cuda.synchronize()
print(f'host {1e3 * host / 6:.2f} ms/step (queue drained between steps); wall {1e3 * (time.perf_counter() - t0) / 6:.2f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(6):
    step.set_input(a, b, z, c); step.optimize_parameters()
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumtime').print_stats(40)
