#!/usr/bin/env python3
"""Generate golden input/output vectors from the *actual* reference (zhiyuns/AFCM).

Run ONLY in the build container where /root/reference is mounted:

    python tools/gen_golden.py            # writes tests/golden/*.npz

The reference is imported (never copied) through the namespace-package bypass of
SURVEY.md Appendix B: `models/__init__.py` pulls torchvision/fvcore which are absent, so a
bare `models` namespace module is pre-registered.  On CPU tensors every reference op takes
its `impl='ref'` aten path (SG3OPS/filtered_lrelu.py:114-116, upfirdn2d.py:160-162,
bias_act.py:84-86), which is the specification the HIP kernels are held to.

The fixtures are data only: inputs, parameters, outputs, and gradients of
``sum(y * r)`` for a fixed random ``r``.  Nothing from the reference's source travels.
"""
import os
import re
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    pkg = types.ModuleType('models')
    pkg.__path__ = [os.path.join(REF, 'models')]
    sys.modules['models'] = pkg
    import torch  # noqa
    import models.networks.stylegan3.networks_stylegan3 as net
    from models.networks.stylegan3.torch_utils.ops import filtered_lrelu, upfirdn2d, bias_act
    from models.networks.CoModGAN.layers import Conv2dLayer
    return net, filtered_lrelu, upfirdn2d, bias_act, Conv2dLayer


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def save(name, **arrays):
    arrays = {k: v for k, v in arrays.items() if v is not None}
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
    size = os.path.getsize(os.path.join(OUT, name + '.npz'))
    desc = ' '.join(f'{k}{list(np.shape(v))}' for k, v in arrays.items() if '/' not in k)
    print(f'{name:28s} {size/1024:8.1f} KiB  {desc}')


def main():
    import torch
    net, flr, ufd, bact, Conv2dLayer = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(1234)
    design = net.SynthesisLayer.design_lowpass_filter

    # ---------------------------------------------------------------- filtered_lrelu F1..F8
    def flrelu_case(name, xshape, fu, fd, up, down, padding, gain=float(np.sqrt(2)), slope=0.2, clamp=256.0,
                    flip_filter=False, xscale=1.0, bias=True):
        x = (torch.randn(xshape) * xscale).requires_grad_(True)
        b = (torch.randn(xshape[1]) * 0.5).requires_grad_(True) if bias else None
        y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=padding, gain=gain, slope=slope,
                               clamp=clamp, flip_filter=flip_filter, impl='ref')
        r = torch.randn_like(y)
        gs = torch.autograd.grad((y * r).sum(), [x] + ([b] if bias else []))
        save(name, x=_np(x), b=_np(b), fu=_np(fu), fd=_np(fd), y=_np(y), r=_np(r), dx=_np(gs[0]),
             db=_np(gs[1]) if bias else None,
             meta=np.array([up, down] + list(padding), dtype=np.int64),
             fmeta=np.array([gain, slope, -1.0 if clamp is None else clamp, float(flip_filter)], dtype=np.float64))

    # Filters exactly as the layers design them (NET:389-392): e.g. enc0 of the 256^2 model.
    fu12 = design(numtaps=12, cutoff=64.0, width=2 * (181.02 - 64.0), fs=512)
    fd12 = design(numtaps=12, cutoff=56.0, width=2 * (160.0 - 56.0), fs=512)
    fd24 = design(numtaps=24, cutoff=20.0, width=2 * (64.0 - 20.0), fs=512)
    fu24 = design(numtaps=24, cutoff=20.0, width=2 * (64.0 - 20.0), fs=512)
    flrelu_case('F1_up2_down2', [2, 3, 20, 20], fu12, fd12, 2, 2, [9, 8, 9, 8])
    flrelu_case('F2_up2_down4', [1, 2, 38, 38], fu12, fd24, 2, 4, [34, 33, 34, 33])
    flrelu_case('F3_up4_down2', [1, 2, 38, 38], fu24, fd12, 4, 2, [-6, -9, -6, -9])
    flrelu_case('F4_crop', [1, 2, 40, 40], fu12, fd12, 2, 2, [-11, -12, -11, -12])
    flrelu_case('F5_identity', [2, 3, 17, 19], None, None, 1, 1, [0, 0, 0, 0], gain=1.0, slope=1.0)
    flrelu_case('F6_clamp', [1, 2, 20, 20], fu12, fd12, 2, 2, [9, 8, 9, 8], xscale=300.0)
    fa = torch.randn(12); fb = torch.randn(12)
    flrelu_case('F7_flip_asym', [1, 2, 21, 23], fa / fa.abs().sum(), fb / fb.abs().sum(), 2, 2, [9, 8, 7, 10], flip_filter=True)
    flrelu_case('F7b_noflip_asym', [1, 2, 21, 23], fa / fa.abs().sum(), fb / fb.abs().sum(), 2, 2, [9, 8, 7, 10], flip_filter=False)
    fr = design(numtaps=12, cutoff=56.0, width=2 * (160.0 - 56.0), fs=512, radial=True)
    flrelu_case('F8_radial2d', [1, 2, 20, 20], fu12, fr, 2, 2, [9, 8, 9, 8])
    flrelu_case('F9_nobias_noclamp', [1, 1, 24, 18], fu12, fd12, 2, 2, [9, 8, 9, 8], clamp=None, bias=False)
    # Wider plane that spans several kernel tiles in both directions (multi-tile seams).
    flrelu_case('F10_multitile', [1, 1, 150, 150], fu12, fd12, 2, 2, [9, 8, 9, 8])
    flrelu_case('F11_multitile_d4', [1, 1, 150, 150], fu12, fd24, 2, 4, [34, 33, 34, 33])
    flrelu_case('F12_multitile_u4', [1, 1, 86, 86], fu24, fd12, 4, 2, [-6, -9, -6, -9])

    # ---------------------------------------------------------------- upfirdn2d U1..U5
    def ufd_case(name, xshape, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0, fn='upfirdn2d'):
        x = torch.randn(xshape).requires_grad_(True)
        y = getattr(ufd, fn)(x, f, **({'up': up} if fn in ('upfirdn2d', 'upsample2d') else {}),
                             **({'down': down} if fn in ('upfirdn2d', 'downsample2d') else {}),
                             padding=padding, flip_filter=flip_filter, gain=gain, impl='ref')
        r = torch.randn_like(y)
        dx, = torch.autograd.grad((y * r).sum(), [x])
        pad = [padding] * 4 if isinstance(padding, int) else (list(padding) if len(padding) == 4 else [padding[0]] * 2 + [padding[1]] * 2)
        save(name, x=_np(x), f=_np(f), y=_np(y), r=_np(r), dx=_np(dx),
             meta=np.array([up, down] + pad + [int(flip_filter)], dtype=np.int64), fmeta=np.array([gain]),
             fn=np.array(fn))

    f4 = ufd.setup_filter([1, 3, 3, 1])            # 2-D 4x4 (D's resample filter, CoModGAN layers.py:125)
    ufd_case('U1_up2_2d', [2, 3, 16, 16], f4, up=2, padding=[2, 1, 2, 1], gain=4.0)
    ufd_case('U2_down2_2d', [2, 3, 16, 16], f4, down=2, padding=[1, 1, 1, 1])
    sigma = 10.0
    bs = np.floor(sigma * 3)
    g = torch.arange(-bs, bs + 1).div(sigma).square().neg().exp2()   # stylegan3_model.py:25-28
    ufd_case('U3_gauss61_filter2d', [1, 2, 72, 80], g / g.sum(), fn='filter2d')
    ufd_case('U4_negpad_sep', [1, 2, 30, 30], fu12, up=2, down=1, padding=[-3, -2, 4, -5], gain=4.0)
    fasym = torch.randn(5, 7)
    ufd_case('U5_asym2d_flip', [1, 2, 12, 14], fasym, up=2, down=3, padding=[3, 2, 1, 4], flip_filter=True, gain=2.0)
    ufd_case('U6_upsample2d', [1, 2, 12, 14], f4, up=2, fn='upsample2d')
    ufd_case('U7_downsample2d', [1, 2, 12, 14], f4, down=2, fn='downsample2d')

    # ---------------------------------------------------------------- bias_act B1..B4
    def ba_case(name, xshape, dim, act, alpha=None, gain=None, clamp=None, bias=True):
        x = (torch.randn(xshape) * 2).requires_grad_(True)
        b = torch.randn(xshape[dim]).requires_grad_(True) if bias else None
        y = bact.bias_act(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp, impl='ref')
        r = torch.randn_like(y)
        gs = torch.autograd.grad((y * r).sum(), [x] + ([b] if bias else []))
        save(name, x=_np(x), b=_np(b), y=_np(y), r=_np(r), dx=_np(gs[0]), db=_np(gs[1]) if bias else None,
             act=np.array(act), dim=np.array(dim),
             fmeta=np.array([np.nan if v is None else v for v in (alpha, gain, clamp)], dtype=np.float64))

    ba_case('B1_linear_bias_nc', [5, 7], 1, 'linear')
    ba_case('B2_lrelu_clamp_nchw', [2, 3, 9, 11], 1, 'lrelu', gain=float(np.sqrt(2)), clamp=1.5)
    ba_case('B3_lrelu_fc', [4, 32], 1, 'lrelu')
    for act in ['relu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']:
        ba_case(f'B4_{act}', [2, 3, 5, 6], 1, act, clamp=(0.9 if act in ('tanh', 'swish') else None))

    # ---------------------------------------------------------------- modulated_conv2d M1..M3
    def mc_case(name, N, I, O, k, H, W, demodulate, padding, gain=None):
        x = torch.randn(N, I, H, W).requires_grad_(True)
        w = torch.randn(O, I, k, k).requires_grad_(True)
        s = (torch.randn(N, I) + 1).requires_grad_(True)
        ig = None if gain is None else torch.tensor(gain)
        y = net.modulated_conv2d(x=x, w=w, s=s, demodulate=demodulate, padding=padding, input_gain=ig)
        r = torch.randn_like(y)
        dx, dw, ds = torch.autograd.grad((y * r).sum(), [x, w, s])
        save(name, x=_np(x), w=_np(w), s=_np(s), y=_np(y), r=_np(r), dx=_np(dx), dw=_np(dw), ds=_np(ds),
             meta=np.array([int(demodulate), padding], dtype=np.int64),
             fmeta=np.array([np.nan if gain is None else gain]))

    mc_case('M1_demod_k3', 2, 5, 7, 3, 12, 10, True, 2, gain=1.25)
    mc_case('M2_nodemod_k1', 2, 6, 1, 1, 9, 9, False, 0)
    mc_case('M3_demod_k3_wide', 2, 37, 21, 3, 14, 14, True, 2, gain=1.0)

    # ---------------------------------------------------------------- tiny generators G1 (128^2), G2 (256^2)
    def gen_case(name, res, batch, grads):
        torch.manual_seed(res)
        kw = dict(z_dim=32, c_dim=1, w_dim=32, img_resolution=res, img_channels_in=4, img_channels_out=1,
                  mapping_kwargs=dict(num_layers=2),
                  synthesis_kwargs=dict(channel_base=256, channel_max=8, num_layers=14, num_critical=2, first_cutoff=2,
                                        first_stopband=2 ** 2.1, last_stopband_rel=2 ** 0.3, margin_size=10,
                                        output_scale=0.25, skip_resolution=128, conv_kernel=3, filter_size=6,
                                        lrelu_upsampling=2, use_radial_filters=False, conv_clamp=256,
                                        magnitude_ema_beta=0.5 ** (16 / 20e3), cond_mod=True))
        G = net.Stylegan3Generator(**kw).eval()
        # Non-trivial biases / magnitude EMAs so that every term of the layer is exercised.
        with torch.no_grad():
            for n, p in G.named_parameters():
                if n.endswith('.bias') and 'affine' not in n:
                    p.add_(torch.randn_like(p) * 0.1)
        z = torch.randn(batch, 32); c = torch.rand(batch, 1)
        xin = torch.randn(batch, 4, res, res).clamp(-1, 1)
        feats = {}
        hooks = []
        for lname, mod in G.synthesis.named_children():
            if hasattr(mod, 'up_factor'):
                hooks.append(mod.register_forward_hook(lambda m, i, o, lname=lname: feats.__setitem__(lname, o.detach())))
        y = G(z, c, xin)
        for h in hooks:
            h.remove()
        r = torch.randn_like(y)
        pnames = [n for n, _ in G.named_parameters()]
        allg = torch.autograd.grad((y * r).sum(), list(G.parameters()), allow_unused=True)
        gd = {n: g for n, g in zip(pnames, allg) if g is not None}
        arrays = {'sd/' + k: _np(v) for k, v in G.state_dict().items()}
        arrays.update(z=_np(z), c=_np(c), x=_np(xin), y=_np(y), r=_np(r))
        # layer statistics (mean, std, absmax) for every resampling layer; full tensors for a few
        stats = {k: np.array([v.mean().item(), v.std().item(), v.abs().max().item()]) for k, v in feats.items()}
        arrays.update({'stat/' + k: v for k, v in stats.items()})
        for pat in grads:
            for k in gd:
                if re.fullmatch(pat, k):
                    arrays['grad/' + k] = _np(gd[k])
        arrays['gradnorm_names'] = np.array(sorted(gd.keys()))
        arrays['gradnorm'] = np.array([gd[k].norm().item() for k in sorted(gd.keys())])
        arrays['layer_names'] = np.array(list(feats.keys()))
        save(name, **arrays)

    gen_case('G1_tiny128', 128, 2, [r'synthesis\.encoder_0\.weight', r'synthesis\.encoder_5\.bias', r'synthesis\.L3_52_8\.weight',
                                    r'synthesis\.L3_52_8\.affine\.weight', r'synthesis\.L13_128_2\.bias', r'mapping\.fc0\.weight',
                                    r'synthesis\.fc_in\.weight', r'synthesis\.e_16x16\.weight'])
    gen_case('G2_tiny256', 256, 1, [r'synthesis\.encoder_0\.weight', r'synthesis\.encoder_4\.bias', r'synthesis\.L10_.*\.weight',
                                    r'synthesis\.L14_.*\.weight', r'mapping\.embed\.weight'])

    # ---------------------------------------------------------------- layer geometry table of the full-width 256^2 model
    torch.manual_seed(0)
    rows = []
    names = []
    # Constructing the full-width generator only evaluates __init__ (58.5 M randn), no forward.
    Gfull = net.Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1,
                                   mapping_kwargs=dict(num_layers=8),
                                   synthesis_kwargs=dict(channel_base=16384, channel_max=512, num_layers=14, num_critical=2,
                                                         margin_size=10, output_scale=0.25, skip_resolution=128, conv_kernel=3,
                                                         filter_size=6, lrelu_upsampling=2, use_radial_filters=False,
                                                         conv_clamp=256, magnitude_ema_beta=0.5 ** (16 / 20e3), cond_mod=True))
    filt = {}
    for lname, m in Gfull.synthesis.named_children():
        if hasattr(m, 'up_factor'):
            names.append(lname)
            rows.append([m.in_channels, m.out_channels, int(m.in_size[0]), int(m.out_size[0]), m.up_factor, m.down_factor,
                         m.up_taps, m.down_taps] + list(m.padding) + [m.conv_kernel])
            if m.up_filter is not None:
                filt['fu/' + lname] = _np(m.up_filter)
            if m.down_filter is not None:
                filt['fd/' + lname] = _np(m.down_filter)
    save('T256_layer_table', names=np.array(names), table=np.array(rows, dtype=np.int64),
         nparams=np.array(sum(p.numel() for p in Gfull.parameters())),
         sd_keys=np.array(list(Gfull.state_dict().keys())), **filt)


if __name__ == '__main__':
    main()
