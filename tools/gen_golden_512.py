#!/usr/bin/env python3
"""Golden vector G3_tiny512 for the 512^2 variant of the generator (BASELINE config 5: `model.G.img_resolution 512`), captured
from the REAL reference running on CPU in the build container (same import recipe as tools/gen_golden.py; never runs on
the GPU box, the reference does not travel).  Narrow channels (channel_base 1024, channel_max 8) keep the CPU run and the fixture small; the
plane sizes -- 36/52/84/148/276/532, encoder down-4 layers enc4/6/8/10/12, decoder up-4 layers L2/L4/L6/L8/L10 (SURVEY.md
section 8) -- are those of the full-width 512^2 model.  Inputs are stored compactly: x as the uint8 slice values v
(x = 2 v / 255 - 1, data/augment/transforms.py:604-616), the cotangent r as int8 signs."""
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402


def main():
    import torch
    net, *_ = gg._import_reference()
    res = 512
    torch.manual_seed(res)
    kw = dict(z_dim=32, c_dim=1, w_dim=32, img_resolution=res, img_channels_in=4, img_channels_out=1,
              mapping_kwargs=dict(num_layers=2),
              synthesis_kwargs=dict(channel_base=1024, channel_max=8, num_layers=14, num_critical=2, first_cutoff=2,
                                    first_stopband=2 ** 2.1, last_stopband_rel=2 ** 0.3, margin_size=10,
                                    output_scale=0.25, skip_resolution=128, conv_kernel=3, filter_size=6,
                                    lrelu_upsampling=2, use_radial_filters=False, conv_clamp=256,
                                    magnitude_ema_beta=0.5 ** (16 / 20e3), cond_mod=True))
    G = net.Stylegan3Generator(**kw).eval()
    with torch.no_grad():
        for n, p in G.named_parameters():
            if n.endswith('.bias') and 'affine' not in n:
                p.add_(torch.randn_like(p) * 0.1)
    z = torch.randn(1, 32)
    c = torch.rand(1, 1)
    # smooth field quantised to uint8, as the slices of the H5 files are (data/prepare_h5.py:39-41)
    field = torch.nn.functional.interpolate(torch.randn(1, 4, 32, 32), size=(res, res), mode='bicubic', align_corners=False)
    v = ((field - field.min()) / (field.max() - field.min()) * 255).round().clamp(0, 255).to(torch.uint8)
    xin = v.float() * (2.0 / 255.0) - 1.0
    feats = {}
    hooks = [mod.register_forward_hook(lambda m, i, o, lname=lname: feats.__setitem__(lname, o.detach()))
             for lname, mod in G.synthesis.named_children() if hasattr(mod, 'up_factor')]
    y = G(z, c, xin)
    for h in hooks:
        h.remove()
    r8 = (torch.randint(0, 2, y.shape) * 2 - 1).to(torch.int8)
    pnames = [n for n, _ in G.named_parameters()]
    allg = torch.autograd.grad((y * r8.float()).sum(), list(G.parameters()), allow_unused=True)
    gd = {n: g for n, g in zip(pnames, allg) if g is not None}
    arrays = {'sd/' + k: gg._np(v_) for k, v_ in G.state_dict().items()}
    arrays.update(z=gg._np(z), c=gg._np(c), x_u8=gg._np(v), y=gg._np(y), r_i8=gg._np(r8))
    arrays.update({'stat/' + k: np.array([t.mean().item(), t.std().item(), t.abs().max().item()]) for k, t in feats.items()})
    arrays.update({'shape/' + k: np.array(t.shape) for k, t in feats.items()})
    for pat in (r'synthesis\.encoder_0\.weight', r'synthesis\.encoder_4\.bias', r'synthesis\.L2_.*\.weight', r'synthesis\.L14_.*\.weight'):
        for k in gd:
            if re.fullmatch(pat, k):
                arrays['grad/' + k] = gg._np(gd[k])
    arrays['gradnorm_names'] = np.array(sorted(gd.keys()))
    arrays['gradnorm'] = np.array([gd[k].norm().item() for k in sorted(gd.keys())])
    arrays['layer_names'] = np.array(list(feats.keys()))
    gg.save('G3_tiny512', **arrays)
    print({k: tuple(t.shape[-2:]) for k, t in feats.items()})


if __name__ == '__main__':
    main()
