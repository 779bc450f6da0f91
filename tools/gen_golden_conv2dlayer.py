#!/usr/bin/env python3
"""Golden vectors for the resampling Conv2dLayer (models/networks/CoModGAN/layers.py:115-162 on
CoModGAN/torch_utils/ops/conv2d_resample.py:57-155), captured from the actual reference.

Run ONLY in the build container where /root/reference is mounted:   python tools/gen_golden_conv2dlayer.py
Same import recipe as tools/gen_golden_disc.py.  Fixtures are data only: weights, bias, input, cotangent, output and the three
gradients, for every (kernel, up, down) branch of the decomposition."""
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
    from models.networks.CoModGAN.layers import Conv2dLayer
    out = {}
    cases = [(3, 2, 1), (3, 1, 2), (1, 2, 1), (1, 1, 2), (3, 1, 1)]
    for n, (k, up, down) in enumerate(cases):
        torch.manual_seed(100 + n)
        layer = Conv2dLayer(3, 5, kernel_size=k, up=up, down=down, activation='lrelu', conv_clamp=(2.0 if n == 0 else None))
        with torch.no_grad():
            layer.bias.add_(torch.randn(5) * 0.3)
        x = torch.randn(2, 3, 12, 14, requires_grad=True)
        y = layer(x, gain=0.7)
        r = torch.randn_like(y)
        gx, gw, gb = torch.autograd.grad((y * r).sum(), [x, layer.weight, layer.bias])
        for key, v in (('w', layer.weight), ('b', layer.bias), ('x', x), ('y', y), ('r', r), ('gx', gx), ('gw', gw), ('gb', gb)):
            out[f'{n}/{key}'] = v.detach().numpy()
    out['cases'] = np.array(cases, dtype=np.int64)
    path = os.path.join(OUT, 'C1_conv2dlayer_resample.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
