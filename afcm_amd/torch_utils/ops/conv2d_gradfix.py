"""conv2d_gradfix: the reference routes every plain convolution through
``conv2d_gradfix.conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1)``
(SG3OPS/conv2d_gradfix.py:37-40), which on torch >= 1.11 is a plain ``F.conv2d`` (:53-55).
Here the same entry point runs the MFMA convolution kernels; the generator only needs
stride 1, dilation 1, groups 1 and square 1x1 / 3x3 kernels, anything else raises."""
import contextlib

import torch

from . import conv2d as _conv

enabled = True                      # kept for API compatibility (SG3OPS/conv2d_gradfix.py:21)
weight_gradients_disabled = False   # see no_weight_gradients()


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    """API-compatible with SG3OPS/conv2d_gradfix.py:24-32; weight gradients are only computed when autograd asks."""
    global weight_gradients_disabled
    old = weight_gradients_disabled
    if disable:
        weight_gradients_disabled = True
    yield
    weight_gradients_disabled = old


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if stride not in (1, (1, 1)) or dilation not in (1, (1, 1)) or groups != 1:
        raise NotImplementedError('afcm_amd conv2d supports stride 1, dilation 1, groups 1 (all the generator uses)')
    if isinstance(padding, (list, tuple)):
        if padding[0] != padding[1]:
            raise NotImplementedError('afcm_amd conv2d needs equal padding in both directions')
        padding = padding[0]
    w = weight.detach() if weight_gradients_disabled else weight
    y = _conv.scaled_conv2d(input, w, None, None, int(padding))
    if bias is not None:
        y = y + bias.to(y.dtype).reshape(1, -1, 1, 1)
    return y
