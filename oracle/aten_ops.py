"""Plain-aten CPU restatement of the reference's fused-op semantics (oracle; test-only).

Each function states which reference lines it follows.  Shorthand:
SG3OPS = /root/reference/models/networks/stylegan3/torch_utils/ops,
NET    = /root/reference/models/networks/stylegan3/networks_stylegan3.py.

Everything here is ordinary differentiable torch code, so gradients for parity tests come
from autograd on this restatement.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)

# name -> (fn(x, alpha), default alpha, default gain)   [SG3OPS/bias_act.py:21-31]
ACTIVATIONS = {
    'linear':   (lambda x, a: x,                       0.0, 1.0),
    'relu':     (lambda x, a: F.relu(x),               0.0, SQRT2),
    'lrelu':    (lambda x, a: F.leaky_relu(x, a),      0.2, SQRT2),
    'tanh':     (lambda x, a: torch.tanh(x),           0.0, 1.0),
    'sigmoid':  (lambda x, a: torch.sigmoid(x),        0.0, 1.0),
    'elu':      (lambda x, a: F.elu(x),                0.0, 1.0),
    'selu':     (lambda x, a: F.selu(x),               0.0, 1.0),
    'softplus': (lambda x, a: F.softplus(x),           0.0, 1.0),
    'swish':    (lambda x, a: torch.sigmoid(x) * x,    0.0, SQRT2),
}


def _pad4(padding):
    """int | [px, py] | [px0, px1, py0, py1] -> (px0, px1, py0, py1).  [SG3OPS/upfirdn2d.py:44-53]"""
    if isinstance(padding, (int, np.integer)):
        padding = [int(padding)] * 2
    padding = [int(p) for p in padding]
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    assert len(padding) == 4
    return tuple(padding)


def _pair(v):
    if isinstance(v, (int, np.integer)):
        return int(v), int(v)
    a, b = v
    return int(a), int(b)


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """y = clamp(act(x + b) * gain).  Follows _bias_act_ref, SG3OPS/bias_act.py:91-120."""
    fn, def_alpha, def_gain = ACTIVATIONS[act]
    alpha = def_alpha if alpha is None else float(alpha)
    gain = def_gain if gain is None else float(gain)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = fn(x, alpha)
    if gain != 1:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Zero-insert upsample, pad/crop, FIR, decimate.  Follows _upfirdn2d_ref, SG3OPS/upfirdn2d.py:167-211.

    1-D `f` is applied separably (along W, then along H); the gain is split as
    gain**(ndim/2) per pass (line 196).  The FIR is a true convolution unless `flip_filter`
    (line 198-199: aten's conv is a correlation, so the taps are reversed when flip_filter is False).
    """
    assert x.ndim == 4
    n, c, h, w = x.shape
    upx, upy = _pair(up)
    dnx, dny = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    assert f.ndim in (1, 2)

    # zero insertion: sample (i, j) lands on (i*upy, j*upx), zeros follow it       [lines 187-189]
    z = x.new_zeros([n, c, h * upy, w * upx])
    z[:, :, ::upy, ::upx] = x
    # F.pad crops for negative amounts, which is exactly lines 192-193
    z = F.pad(z, [px0, px1, py0, py1])

    taps = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:
        taps = taps.flip(list(range(taps.ndim)))
    if taps.ndim == 2:
        k = taps[None, None].expand(c, 1, *taps.shape)
        z = F.conv2d(z, k, groups=c)
    else:
        z = F.conv2d(z, taps.reshape(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)
        z = F.conv2d(z, taps.reshape(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    return z[:, :, ::dny, ::dnx]                                                   # [line 210]


def _fsize(f):
    if f is None:
        return 1, 1
    return int(f.shape[-1]), int(f.shape[0])


def filter2d(x, f, padding=0, flip_filter=False, gain=1):
    """Same-size FIR.  Padding rule of SG3OPS/upfirdn2d.py:301-309."""
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _fsize(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1):
    """SG3OPS/upfirdn2d.py:339-348."""
    upx, upy = _pair(up)
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _fsize(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1):
    """SG3OPS/upfirdn2d.py:378-387."""
    dx, dy = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _fsize(f)
    p = [px0 + (fw - dx + 1) // 2, px1 + (fw - dx) // 2, py0 + (fh - dy + 1) // 2, py1 + (fh - dy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=SQRT2, slope=0.2, clamp=None,
                   flip_filter=False, codes=None, record=None):
    """bias -> upfirdn(up, gain up^2) -> leaky-ReLU * gain -> clamp -> upfirdn(down).

    Follows _filtered_lrelu_ref, SG3OPS/filtered_lrelu.py:121-153.

    Test instrumentation (not part of the reference signature): ``record`` (a list) receives the pre-activation tensor on the
    upsampled grid; ``codes`` (uint8 [N, C, >= rows used, >= cols], bit 0 = negative branch, bit 1 = clamped -- the meaning of the
    plugin's 2-bit codes, SG3OPS/filtered_lrelu.cu:494-505) imposes the branch decisions of another implementation, so that a
    gradient comparison can separate "a pre-activation within rounding distance of 0 took the other branch" from real errors.
    """
    px0, px1, py0, py1 = _pad4(padding)
    fuw, fuh = _fsize(fu)
    fdw, fdh = _fsize(fd)
    n, c, h, w = x.shape
    ow = (w * up + px0 + px1 - (fuw - 1) - (fdw - 1) + (down - 1)) // down        # [line 141]
    oh = (h * up + py0 + py1 - (fuh - 1) - (fdh - 1) + (down - 1)) // down        # [line 142]
    y = bias_act(x, b)
    y = upfirdn2d(y, fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if record is not None:
        record.append(y.detach())
    if codes is None:
        y = bias_act(y, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    else:
        full = torch.zeros(y.shape, dtype=torch.uint8)
        hh, ww = min(y.shape[2], codes.shape[2]), min(y.shape[3], codes.shape[3])
        full[:, :, :hh, :ww] = codes[:, :, :hh, :ww]
        if hh < y.shape[2] or ww < y.shape[3]:      # rows / columns the other implementation never stored: own decision
            own = (y.detach() < 0).to(torch.uint8)
            own[:, :, :hh, :ww] = full[:, :, :hh, :ww]
            full = own
        v = y * torch.where((full & 1) != 0, torch.full_like(y, slope), torch.ones_like(y)) * gain
        if clamp is not None and clamp >= 0:
            v = torch.where((full & 2) != 0, v.detach().clamp(-clamp, clamp), v)
        y = v
    y = upfirdn2d(y, fd, down=down, flip_filter=flip_filter)
    assert tuple(y.shape) == (n, c, oh, ow), (y.shape, (n, c, oh, ow))
    return y


def modulated_conv2d(x, w, s, demodulate=True, padding=0, input_gain=None):
    """Per-sample weight (de)modulated convolution, literal grouped-conv form.  Follows NET:25-64."""
    n = x.shape[0]
    o, i, kh, kw = w.shape
    if demodulate:
        w = w * w.square().mean([1, 2, 3], keepdim=True).rsqrt()                   # [line 42]
        s = s * s.square().mean().rsqrt()                                          # [line 43]  (whole batch)
    wn = w[None] * s[:, None, :, None, None]                                       # [N,O,I,k,k]  [line 46-47]
    if demodulate:
        d = (wn.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()                        # [line 51]
        wn = wn * d[:, :, None, None, None]
    if input_gain is not None:
        wn = wn * input_gain.expand(n, i)[:, None, :, None, None]                  # [line 55-57]
    y = F.conv2d(x.reshape(1, n * i, *x.shape[2:]), wn.reshape(n * o, i, kh, kw).to(x.dtype), padding=padding, groups=n)
    return y.reshape(n, o, *y.shape[2:])


def conv2d(x, w, padding=0):
    """conv2d_gradfix.conv2d on torch >= 1.11 is plain F.conv2d (SG3OPS/conv2d_gradfix.py:37-58)."""
    return F.conv2d(x, w.to(x.dtype), padding=padding)


def psnr(pred, target):
    """PSNR as the reference evaluates it: map [-1,1] -> [0,1], clip (train.py:93-96), normalise each
    image by its own maximum and take 10*log10(1/MSE) (util/evaluation.py:31-37)."""
    p = ((pred + 1) / 2).clamp(0, 1).double()
    t = ((target + 1) / 2).clamp(0, 1).double()
    out = []
    for a, b in zip(p, t):
        a = a / a.max().clamp_min(1e-12)
        b = b / b.max().clamp_min(1e-12)
        mse = (a - b).square().mean().clamp_min(1e-20)
        out.append(10.0 * math.log10(1.0 / mse.item()))
    return float(np.mean(out))
