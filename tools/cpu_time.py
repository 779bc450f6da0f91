import sys, time, torch
sys.path.insert(0, '.')
from afcm_amd import layer_schedule as sched, synthetic
from afcm_amd.networks_stylegan3 import Stylegan3Generator
from afcm_amd.stylegan3_model import StyleGAN3GeneratorStep
dev = torch.device('cuda', 0)
kw = dict(sched.DEFAULT_SYNTHESIS_KWARGS)
G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1,
                       mapping_kwargs=dict(num_layers=8), synthesis_kwargs=dict(kw, compute_dtype=torch.bfloat16)).to(dev).train()
step = StyleGAN3GeneratorStep(G, lr_G=0.0025, lambda_L1=100.0)
a, b, z, c = synthetic.generator_inputs(16, size=256, seed=0, device=dev)
for _ in range(3):
    step.set_input(a, b, z, c); step.optimize_parameters()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    step.set_input(a, b, z, c); step.optimize_parameters()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'enqueue {1e3*(t1-t0)/6:.1f} ms/step, total {1e3*(t2-t0)/6:.1f} ms/step')
