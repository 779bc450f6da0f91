"""upfirdn2d: pad, zero-insert upsample, FIR-filter and decimate batches of 2-D images on the GPU.

Drop-in for the reference module SG3OPS/upfirdn2d.py: ``setup_filter`` (:70), ``upfirdn2d`` (:118),
``filter2d`` (:277), ``upsample2d`` (:313), ``downsample2d`` (:352) keep their signatures and defaults.
``impl='cuda'`` on a ROCm tensor runs the HIP kernel ``afcm_upfirdn2d``; the backward pass is the op
itself with up/down swapped and the filter flipped (SG3OPS/upfirdn2d.py:250-269), so gradients of
any order are available.
"""
import numpy as np
import torch

from ... import _lib


def _parse_scaling(scaling):
    if isinstance(scaling, (int, np.integer)):
        scaling = [int(scaling)] * 2
    assert isinstance(scaling, (list, tuple)) and len(scaling) == 2
    sx, sy = (int(v) for v in scaling)
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    if isinstance(padding, (int, np.integer)):
        padding = [int(padding)] * 2
    assert isinstance(padding, (list, tuple))
    padding = [int(v) for v in padding]
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    assert len(padding) == 4
    return tuple(padding)


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Build a float32 FIR filter tensor: `[taps]` (separable) or `[fh, fw]`.  Behaviour of SG3OPS/upfirdn2d.py:70-114."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def _launch(x, f2d, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain):
    """One call of the C ABI `afcm_upfirdn2d` with a 2-D `[fh, fw]` filter (replaces `_plugin.upfirdn2d`)."""
    lib = _lib.load()
    n, c, xh, xw = x.shape
    fh, fw = f2d.shape
    yw = (xw * upx + px0 + px1 - fw + downx) // downx
    yh = (xh * upy + py0 + py1 - fh + downy) // downy
    if yw < 1 or yh < 1:
        raise RuntimeError('upfirdn2d: output must be at least 1x1')
    y = torch.empty([n, c, yh, yw], dtype=x.dtype, device=x.device)
    rc = lib.afcm_upfirdn2d(_lib.ptr(y), _lib.ptr(x), _lib.ptr(f2d), _lib.dtype_code(x), n, c, xh, xw, yh, yw, fh, fw,
                            upx, upy, downx, downy, px0, py0, int(flip), float(gain), _lib.stream_ptr(x))
    if _lib.check(rc, 'upfirdn2d') == _lib.E_NOKERNEL:
        raise RuntimeError(f'upfirdn2d: filters larger than 4096 taps are not supported (got {fh}x{fw})')
    return y


def _forward_raw(x, f, up, down, padding, flip_filter, gain):
    """Un-differentiated op.  A 1-D filter runs as two 1-D passes with gain 1 then `gain` (SG3OPS/upfirdn2d.py:244-245)."""
    _lib.require_gpu(x, f)
    upx, upy = up
    downx, downy = down
    px0, px1, py0, py1 = padding
    if x.ndim != 4:
        raise RuntimeError('x must be rank 4')
    if x.numel() == 0:
        raise RuntimeError('x is empty')
    x = x.contiguous()
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    if f.dtype != torch.float32:
        raise RuntimeError('f must be float32')
    if f.ndim == 1 and f.shape[0] == 1:
        f = f.square().unsqueeze(0)
    f = f.contiguous()
    if f.ndim == 2:
        return _launch(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain)
    y = _launch(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip_filter, 1.0)
    return _launch(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip_filter, gain)


class _Upfirdn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, up, down, padding, flip_filter, gain):
        y = _forward_raw(x, f, up, down, padding, flip_filter, gain)
        ctx.save_for_backward(f)
        ctx.cfg = (up, down, padding, flip_filter, gain, x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        f, = ctx.saved_tensors
        (upx, upy), (downx, downy), (px0, _, py0, _), flip_filter, gain, (_, _, ih, iw) = ctx.cfg
        _, _, oh, ow = dy.shape
        fw, fh = _get_filter_size(f)
        p = (fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _Upfirdn2d.apply(dy, f, (downx, downy), (upx, upy), p, not flip_filter, gain)
        assert not ctx.needs_input_grad[1]
        return dx, None, None, None, None, None, None


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Pad, upsample, filter, downsample.  Same signature and defaults as SG3OPS/upfirdn2d.py:118."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    # impl='ref' on a GPU tensor: there is ONE device implementation of this op, the HIP kernel (the reference's 'ref' twin is
    # its aten composition, SG3OPS/upfirdn2d.py:167-211; reference callers that ask for it still run, on the same kernel).
    # A CPU tensor raises in _launch for either value: this package has no CPU path.
    return _Upfirdn2d.apply(x, f, _parse_scaling(up), _parse_scaling(down), _parse_padding(padding), bool(flip_filter), float(gain))


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Same-size FIR (SG3OPS/upfirdn2d.py:277-309)."""
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Upsampling FIR (SG3OPS/upfirdn2d.py:313-348)."""
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Downsampling FIR (SG3OPS/upfirdn2d.py:352-387)."""
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
