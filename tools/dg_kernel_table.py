#!/usr/bin/env python3
"""Steady-state kernel time of ONE full D + G iteration (bench.py --with-discriminator's workload) by kernel, from torch.profiler after
warm-up iterations (rocprofv3 --stats over the whole process also counts MIOpen's find / benchmark launches of the first iterations).
    python tools/dg_kernel_table.py [--top 40]"""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import layer_schedule as sched, synthetic
from afcm_amd.networks_stylegan3 import Stylegan3Generator
from afcm_amd.networks_discriminator import CoModDiscriminator
from afcm_amd.stylegan3_model import StyleGAN3Step

ap = argparse.ArgumentParser()
ap.add_argument('--top', type=int, default=45)
a = ap.parse_args()
dt = torch.bfloat16
dev = torch.device('cuda', 0)
torch.manual_seed(0)
G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1, mapping_kwargs=dict(num_layers=8),
                       synthesis_kwargs=dict(sched.DEFAULT_SYNTHESIS_KWARGS, compute_dtype=dt)).to(dev).train()
D = CoModDiscriminator(c_dim=0, img_resolution=256, img_channels=5, channel_base=16384, channel_max=512, num_fp16_res=4, conv_clamp=256,
                       block_kwargs=dict(fp16_dtype=dt), epilogue_kwargs=dict(mbstd_group_size=16)).to(dev)
step = StyleGAN3Step(G, D, lr_G=0.0025, lr_D=0.0025, lambda_L1=100.0, lambda_r1=10.0)
real_A, real_B, z, c = synthetic.generator_inputs(16, size=256, seed=0, device=dev)


def one():
    step.set_input(real_A, real_B, z, c)
    step.optimize_parameters()


for _ in range(4):
    one()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    one()
    torch.cuda.synchronize()
ker = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and 'Memcpy' not in e.name and 'Memset' not in e.name]


def dur(e):
    for k in ('device_time', 'cuda_time', 'self_device_time_total'):
        v = getattr(e, k, None)
        if v:
            return float(v)
    return float(e.time_range.elapsed_us())


tot = collections.defaultdict(float); cnt = collections.Counter()
for e in ker:
    n = e.name.split('(')[0][:100]
    tot[n] += dur(e); cnt[n] += 1
total = sum(tot.values())
print(f'# one D + G iteration: {len(ker)} kernel launches, {total / 1e3:.2f} ms of kernel time')


def group(n):
    if 'fwd16x' in n or 'conv2d_fwd' in n: return 'afcm conv (fwd / dgrad)'
    if 'wgrad' in n: return 'afcm weight gradient'
    if 'flrelu' in n: return 'afcm filtered_lrelu'
    if 'upfirdn2d' in n: return 'afcm upfirdn2d'
    if 'bias_act' in n: return 'afcm bias_act'
    if 'plane_dot' in n or 'scale_planes' in n or 'layer_bwd' in n: return 'afcm plane dots / scales'
    if 'afcm' in n: return 'afcm other'
    if 'miopen' in n.lower() or 'igemm' in n or 'naive_conv' in n or 'ck::' in n or '_ZN2ck' in n or 'Sp3Asm' in n or 'Im2' in n or 'Col2' in n: return 'MIOpen convolutions (fp32 blocks)'
    if n.startswith('Cijk'): return 'hipBLASLt / rocBLAS GEMMs'
    return 'framework elementwise / reductions / copies'


g = collections.defaultdict(float); gc = collections.Counter()
for n, v in tot.items():
    g[group(n)] += v; gc[group(n)] += cnt[n]
for k, v in sorted(g.items(), key=lambda kv: -kv[1]):
    print(f'{v / 1e3:8.2f} ms {100 * v / total:5.1f} %  {gc[k]:5d} launches  {k}')
print()
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:a.top]:
    print(f'{v / 1e3:8.3f} ms  {cnt[n]:4d} x {v / cnt[n]:8.1f} us  {n}')
