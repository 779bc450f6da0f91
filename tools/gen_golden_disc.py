#!/usr/bin/env python3
"""Golden vectors for the discriminator path, captured from the *actual* reference
(models/networks/CoModGAN/generator.py:780-836 and the backward_D arithmetic of models/comodgan_model.py:128-149).

Run ONLY in the build container where /root/reference is mounted:   python tools/gen_golden_disc.py
Same import recipe as tools/gen_golden.py.  Fixtures are data only: state dict, inputs, logits,
parameter gradients of the two D loss terms, the R1 penalty and its (double-backward) parameter gradients, and the gradient
the generator receives through D.
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    pkg = types.ModuleType('models')
    pkg.__path__ = [os.path.join(REF, 'models')]
    sys.modules['models'] = pkg
    import torch
    from models.networks.CoModGAN.generator import CoModDiscriminator

    only = sys.argv[1:]
    for name, res, kw, n in [('D1_tiny64', 64, dict(channel_base=512, channel_max=16, epilogue_kwargs=dict(mbstd_group_size=2)), 4),
                             ('D2_tiny128_clamp', 128, dict(channel_base=1024, channel_max=12, conv_clamp=256,
                                                            epilogue_kwargs=dict(mbstd_group_size=2)), 2),
                             # the conditional form of configs/adni/stylegan3/cmsr.yml:13 (model.D.c_dim = 1: the slice fraction)
                             ('D3_tiny64_cond', 64, dict(c_dim=1, channel_base=512, channel_max=16, epilogue_kwargs=dict(mbstd_group_size=2)), 4)]:
        if only and name not in only:
            continue
        torch.manual_seed(77)
        kw = dict(kw)
        c_dim = kw.pop('c_dim', 0)
        D = CoModDiscriminator(c_dim=c_dim, img_resolution=res, img_channels=5, **kw)
        with torch.no_grad():
            for p in D.parameters():                       # biases start at 0: make every term of the arithmetic count
                if p.ndim == 1:
                    p.add_(torch.randn_like(p) * 0.1)
        fake = torch.randn(n, 5, res, res)
        real = torch.randn(n, 5, res, res)
        c = torch.rand(n, c_dim) if c_dim > 0 else None
        out = {'sd/' + k: v.detach().numpy() for k, v in D.state_dict().items()}
        out['fake'], out['real'] = fake.numpy(), real.numpy()
        if c is not None:
            out['c'] = c.numpy()
        params = dict(D.named_parameters())
        names = sorted(params)
        # D step, fake half (comodgan_model.py:133-135)
        gen_logits = D(fake, c)
        loss_fake = torch.nn.functional.softplus(gen_logits).mean()
        g_fake = torch.autograd.grad(loss_fake, [params[k] for k in names])
        out['gen_logits'] = gen_logits.detach().numpy()
        out['loss_fake'] = np.array(loss_fake.item())
        # D step, real half with R1 (comodgan_model.py:137-149), lambda_r1 = 10
        real_tmp = real.detach().requires_grad_(True)
        real_logits = D(real_tmp, c)
        loss_real = torch.nn.functional.softplus(-real_logits).mean()
        r1_grads = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_tmp], create_graph=True, only_inputs=True)[0]
        loss_r1 = r1_grads.square().sum([1, 2, 3]).mean() * 0.5
        g_real = torch.autograd.grad(loss_real + loss_r1 * 10.0, [params[k] for k in names], retain_graph=True)
        # the R1 term on its own (the double backward): inside greal it is only 0-6 % of the magnitude, so a tolerance on the sum
        # would pass an R1 error of tens of percent -- pinned separately.  Parameters the penalty does not
        # reach (e.g. the last bias) have no gradient: stored as zeros.
        g_r1 = torch.autograd.grad(loss_r1, [params[k] for k in names], allow_unused=True)
        out['real_logits'] = real_logits.detach().numpy()
        out['r1_grads'] = r1_grads.detach().numpy()
        out['loss_real'], out['loss_r1'] = np.array(loss_real.item()), np.array(loss_r1.item())
        # G step through D (stylegan3_model.py:93-95): gradient w.r.t. the image
        img = fake.detach().requires_grad_(True)
        loss_g = torch.nn.functional.softplus(-D(img, c)).mean()
        out['g_img'] = torch.autograd.grad(loss_g, img)[0].numpy()
        for k, a, b in zip(names, g_fake, g_real):
            out['gfake/' + k] = a.numpy()
            out['greal/' + k] = b.numpy()
        for k, a in zip(names, g_r1):
            out['gr1/' + k] = (torch.zeros_like(params[k]) if a is None else a).numpy()
        out['names'] = np.array(names)
        out['meta'] = np.array([res, n, kw['channel_base'], kw['channel_max'], kw['epilogue_kwargs']['mbstd_group_size'],
                                int(kw.get('conv_clamp') or -1)], dtype=np.int64)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
        print(name, os.path.getsize(os.path.join(OUT, name + '.npz')) // 1024, 'KiB', 'params', sum(p.numel() for p in D.parameters()),
              'loss_r1', loss_r1.item())


if __name__ == '__main__':
    main()
