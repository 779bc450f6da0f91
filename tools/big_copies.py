#!/usr/bin/env python3
"""Where does one generator training step copy large tensors?  Logs every Tensor.contiguous() that actually copies and every torch.Tensor.to()
/ clone() of more than --min elements with the innermost afcm_amd frames.  (Census aid: tools/launch_census.py names the kernels, not the lines.)"""
import argparse, collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import layer_schedule as sched, synthetic
from afcm_amd.networks_stylegan3 import Stylegan3Generator
from afcm_amd.stylegan3_model import StyleGAN3GeneratorStep
ap = argparse.ArgumentParser(); ap.add_argument('--min', type=int, default=4_000_000); a = ap.parse_args()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1, mapping_kwargs=dict(num_layers=8),
                       synthesis_kwargs=dict(sched.DEFAULT_SYNTHESIS_KWARGS, compute_dtype=torch.bfloat16)).to(dev).train()
step = StyleGAN3GeneratorStep(G, lr_G=0.0025, lambda_L1=100.0)
real_A, real_B, z, c = synthetic.generator_inputs(16, size=256, seed=0, device=dev)
def one():
    step.set_input(real_A, real_B, z, c); step.optimize_parameters()
for _ in range(2): one()
sites = collections.Counter()
def where():
    fr = [f for f in traceback.extract_stack()[:-2] if 'afcm_amd' in f.filename]
    return ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in fr[-3:][::-1])
orig_contig, orig_clone = torch.Tensor.contiguous, torch.Tensor.clone
def contig(self, *k, **kw):
    if not self.is_contiguous() and self.numel() >= a.min: sites[('contiguous', tuple(self.shape), str(self.dtype), where())] += 1
    return orig_contig(self, *k, **kw)
def clone(self, *k, **kw):
    if self.numel() >= a.min: sites[('clone', tuple(self.shape), str(self.dtype), where())] += 1
    return orig_clone(self, *k, **kw)
torch.Tensor.contiguous, torch.Tensor.clone = contig, clone
one(); torch.cuda.synchronize()
for k, v in sites.most_common(): print(v, k)
