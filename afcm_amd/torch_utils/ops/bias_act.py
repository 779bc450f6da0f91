"""bias_act: fused bias + activation + gain + clamp on the GPU.

Drop-in for the reference's ``bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None,
clamp=None, impl='cuda')`` (SG3OPS/bias_act.py:52-86).  ``impl='cuda'`` on a ROCm tensor runs the HIP
kernel ``afcm_bias_act``; first- and second-order gradients re-enter the same kernel in its
grad=1 / grad=2 modes (the plugin contract of SG3OPS/bias_act.py:142-203).
"""
import collections
import math

import torch

from ... import _lib

ActSpec = collections.namedtuple('ActSpec', 'def_alpha def_gain cuda_idx ref has_2nd_grad')

# Same table as SG3OPS/bias_act.py:21-31 (defaults, kernel index, which tensor the gradient needs).
activation_funcs = {
    'linear':   ActSpec(0.0, 1.0,          1, '',  False),
    'relu':     ActSpec(0.0, math.sqrt(2), 2, 'y', False),
    'lrelu':    ActSpec(0.2, math.sqrt(2), 3, 'y', False),
    'tanh':     ActSpec(0.0, 1.0,          4, 'y', True),
    'sigmoid':  ActSpec(0.0, 1.0,          5, 'y', True),
    'elu':      ActSpec(0.0, 1.0,          6, 'y', True),
    'selu':     ActSpec(0.0, 1.0,          7, 'y', True),
    'softplus': ActSpec(0.0, 1.0,          8, 'y', True),
    'swish':    ActSpec(0.0, math.sqrt(2), 9, 'x', True),
}


def _launch(x, b, xref, yref, dy, grad, dim, spec, alpha, gain, clamp):
    """One call of the C ABI `afcm_bias_act` (replaces `_plugin.bias_act`, SG3OPS/bias_act.cpp:32)."""
    _lib.require_gpu(x, b, xref, yref, dy)
    lib = _lib.load()
    y = torch.empty_like(x)
    if x.numel() == 0:
        return y
    nb = 0
    inner = 1
    if b is not None:
        if b.ndim != 1 or b.shape[0] != x.shape[dim]:
            raise RuntimeError('b must be a vector with the same number of elements as x has along dim')
        if b.dtype != x.dtype:
            raise RuntimeError('x and b must have the same dtype')
        nb = b.shape[0]
        inner = int(math.prod(x.shape[dim + 1:]))
    for t in (xref, yref, dy):
        if t is not None and (t.shape != x.shape or t.dtype != x.dtype):
            raise RuntimeError('xref, yref and dy must have the same shape and dtype as x')
    rc = lib.afcm_bias_act(_lib.ptr(y), _lib.ptr(x), _lib.ptr(b), _lib.ptr(xref), _lib.ptr(yref), _lib.ptr(dy),
                           _lib.dtype_code(x), x.numel(), inner, nb, grad, spec.cuda_idx, alpha, gain, clamp, _lib.stream_ptr(x))
    _lib.check(rc, 'bias_act')
    return y


def _opt(t):
    return t if (t is not None and t.numel() > 0) else None


class _SumPlanes(torch.autograd.Function):
    """t.sum([0, 2, 3]) of an NCHW tensor on the per-plane reduction kernel (C ABI afcm_plane_dot; fp32 accumulation, the result
    in t's dtype as torch's own sum): the bias gradients of the discriminator's 60 bias_act calls per iteration were 93 us
    framework reductions each.  Differentiable (the bias gradient of a first backward is part of the R1 graph)."""

    @staticmethod
    def forward(ctx, t):
        from .conv2d import plane_dot
        ctx.shape = tuple(t.shape)
        return plane_dot(t).sum(0).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.reshape(1, -1, 1, 1).expand(ctx.shape)


def _sum_except(t, dim):
    if t.ndim == 4 and dim == 1 and t.is_cuda and t.dtype in (torch.float32, torch.bfloat16, torch.float16) and t.numel() > 0:
        return _SumPlanes.apply(t)
    return t.sum([i for i in range(t.ndim) if i != dim])


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, dim, act, alpha, gain, clamp):
        spec = activation_funcs[act]
        x = x.contiguous()
        b = b.contiguous() if b is not None else None
        y = x
        if act != 'linear' or gain != 1 or clamp >= 0 or b is not None:
            y = _launch(x, b, None, None, None, 0, dim, spec, alpha, gain, clamp)
        need_x = 'x' in spec.ref or spec.has_2nd_grad
        ctx.save_for_backward(x if need_x else None, b if need_x else None, y if 'y' in spec.ref else None)
        ctx.cfg = (dim, act, alpha, gain, clamp)
        return y

    @staticmethod
    def backward(ctx, dy):
        dim, act, alpha, gain, clamp = ctx.cfg
        x, b, y = ctx.saved_tensors
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy.contiguous()
            if act != 'linear' or gain != 1 or clamp >= 0:
                dx = _BiasActGrad.apply(dx, x, b, y, dim, act, alpha, gain, clamp)
        if ctx.needs_input_grad[1]:
            db = _sum_except(dx, dim)
        return dx, db, None, None, None, None, None


class _BiasActGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, dim, act, alpha, gain, clamp):
        spec = activation_funcs[act]
        dx = _launch(dy, b, x, y, None, 1, dim, spec, alpha, gain, clamp)
        ctx.save_for_backward(dy if spec.has_2nd_grad else None, x, b, y)
        ctx.cfg = (dim, act, alpha, gain, clamp)
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dim, act, alpha, gain, clamp = ctx.cfg
        spec = activation_funcs[act]
        dy, x, b, y = ctx.saved_tensors
        d_dx = d_dx.contiguous()
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, dim, act, alpha, gain, clamp)
        if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _launch(d_dx, b, x, y, dy, 2, dim, spec, alpha, gain, clamp)
        if spec.has_2nd_grad and ctx.needs_input_grad[2]:
            d_b = _sum_except(d_x, dim)
        return d_dy, d_x, d_b, None, None, None, None, None, None


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """Fused bias + activation.  Same signature and defaults as SG3OPS/bias_act.py:52."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    # impl='ref' on a GPU tensor runs the same HIP kernel (one device implementation; the reference's 'ref' twin is the aten
    # composition of SG3OPS/bias_act.py:91-120).  CPU tensors raise below for either value: no CPU path in this package.
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clamp = float(clamp if clamp is not None else -1)
    _lib.require_gpu(x, b)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim
        assert b.shape[0] == x.shape[dim]
    return _BiasAct.apply(x, b, dim, act, alpha, gain, clamp)
