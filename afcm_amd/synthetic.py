"""Synthetic MR-like slices for benchmarks and smoke tests: smooth random field x
elliptical 'brain' mask, background exactly -1, quantised to uint8 and mapped by 2*v/255 - 1 as the
reference's Normalize does (data/augment/transforms.py:604-616; data is uint8 0..255, data/prepare_h5.py:39-41)."""
import math

import torch


def mr_like_slices(batch, channels, size, seed=0, device='cpu'):
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(batch, channels, size // 8, size // 8, generator=g)
    field = torch.nn.functional.interpolate(noise, size=(size, size), mode='bicubic', align_corners=False)
    field = (field - field.amin(dim=(2, 3), keepdim=True)) / (field.amax(dim=(2, 3), keepdim=True) - field.amin(dim=(2, 3), keepdim=True) + 1e-6)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing='ij')
    a = 0.75 + 0.1 * torch.rand(batch, 1, 1, 1, generator=g)
    b = 0.85 + 0.1 * torch.rand(batch, 1, 1, 1, generator=g)
    mask = ((xx[None, None] / a) ** 2 + (yy[None, None] / b) ** 2) <= 1.0
    v = torch.where(mask, (field * 0.8 + 0.2) * 255.0, torch.zeros(()))
    v = v.round().clamp(0, 255)
    return (2.0 * v / 255.0 - 1.0).to(device)


def generator_inputs(batch, size=256, z_dim=512, slice_thickness=None, seed=0, device='cpu'):
    """real_A [B,4,H,W], real_B [B,1,H,W], gen_z [B,z_dim], gen_c [B,1] (fractional slice index in [0,1);
    with `slice_thickness` = t the labels are drawn from {0, 1/t, ..., (t-1)/t} as data/cmsr_dataset.py:131-151 does)."""
    g = torch.Generator().manual_seed(seed + 7919)
    real_A = mr_like_slices(batch, 4, size, seed=seed, device=device)
    real_B = mr_like_slices(batch, 1, size, seed=seed + 1, device=device)
    z = torch.randn(batch, z_dim, generator=g).to(device)
    if slice_thickness:
        c = (torch.randint(0, int(slice_thickness), (batch, 1), generator=g).float() / float(slice_thickness)).to(device)
    else:
        c = torch.rand(batch, 1, generator=g).to(device)
    return real_A, real_B, z, c


def psnr(pred, target):
    """PSNR as the reference evaluates it (train.py:93-96 + util/evaluation.py:31-37): [-1,1] -> [0,1], clip,
    per-image max normalisation, 10*log10(1/MSE); averaged over the batch."""
    p = ((pred.double() + 1) / 2).clamp(0, 1)
    t = ((target.double() + 1) / 2).clamp(0, 1)
    vals = []
    for a, b in zip(p, t):
        a = a / a.max().clamp_min(1e-12)
        b = b / b.max().clamp_min(1e-12)
        mse = (a - b).square().mean().clamp_min(1e-20)
        vals.append(10.0 * math.log10(1.0 / mse.item()))
    return sum(vals) / len(vals)
