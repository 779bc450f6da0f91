"""Definition-level numpy restatement (float64) of upfirdn2d / filtered_lrelu (oracle; test-only).

Written from the op *definition* (distilled from
SG3OPS/upfirdn2d.py:167-211 and SG3OPS/filtered_lrelu.py:121-153) with explicit index arithmetic
instead of aten convolutions, so that it shares no code path with ``aten_ops``.  Small inputs only.

Also holds the 2-bit sign/clamp code restatement (SG3OPS/filtered_lrelu.cu:494-505, 567-570):
code bit0 = value was negative before the slope, code==2 = |value| exceeded the clamp.
"""
import numpy as np


def _pad4(p):
    if isinstance(p, (int, np.integer)):
        p = [int(p)] * 2
    p = [int(v) for v in p]
    if len(p) == 2:
        p = [p[0], p[0], p[1], p[1]]
    return p


def _zero_insert_pad(x, up, lo, hi, axis):
    """Along `axis`: put sample i at i*up, zeros elsewhere; then add lo/hi zeros (negative = crop)."""
    x = np.moveaxis(x, axis, -1)
    n = x.shape[-1]
    z = np.zeros(x.shape[:-1] + (n * up,), dtype=x.dtype)
    z[..., ::up] = x
    total = n * up + lo + hi
    out = np.zeros(x.shape[:-1] + (max(total, 0),), dtype=x.dtype)
    # destination index d corresponds to source index d - lo
    d0 = max(lo, 0)
    s0 = max(-lo, 0)
    length = min(n * up - s0, total - d0)
    if length > 0:
        out[..., d0:d0 + length] = z[..., s0:s0 + length]
    return np.moveaxis(out, -1, axis)


def _fir_valid_decimate(z, taps, down, axis):
    """out[o] = sum_k taps[k] * z[o*down + k]   ('valid' correlation, keep every down-th)."""
    z = np.moveaxis(z, axis, -1)
    n = z.shape[-1]
    k = len(taps)
    full = n - k + 1
    assert full >= 1
    nout = (full + down - 1) // down
    out = np.zeros(z.shape[:-1] + (nout,), dtype=np.float64)
    for t in range(k):
        out += taps[t] * z[..., t:t + (nout - 1) * down + 1:down]
    return np.moveaxis(out, -1, axis)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    x = np.asarray(x, dtype=np.float64)
    px0, px1, py0, py1 = _pad4(padding)
    upx = upy = up
    if f is None:
        f = np.ones([1, 1])
    f = np.asarray(f, dtype=np.float64)
    z = _zero_insert_pad(x, upx, px0, px1, axis=3)
    z = _zero_insert_pad(z, upy, py0, py1, axis=2)
    if f.ndim == 1:
        taps = f * np.sqrt(gain)
        if not flip_filter:
            taps = taps[::-1]
        z = _fir_valid_decimate(z, taps, down, axis=3)
        z = _fir_valid_decimate(z, taps, down, axis=2)
        return z
    taps = f * gain
    if not flip_filter:
        taps = taps[::-1, ::-1]
    fh, fw = taps.shape
    H = z.shape[2] - fh + 1
    W = z.shape[3] - fw + 1
    oh = (H + down - 1) // down
    ow = (W + down - 1) // down
    out = np.zeros(z.shape[:2] + (oh, ow))
    for ky in range(fh):
        for kx in range(fw):
            out += taps[ky, kx] * z[:, :, ky:ky + (oh - 1) * down + 1:down, kx:kx + (ow - 1) * down + 1:down]
    return out


def lrelu_codes(u, gain, slope, clamp):
    """Activation on the upsampled grid + the 2-bit code per element.

    u is the up-FIR output *including* the up^2 factor.  Returns (activated, codes uint8).
    """
    v = u * gain
    neg = np.signbit(v)
    v = np.where(neg, v * slope, v)
    code = neg.astype(np.uint8)
    if clamp is not None and clamp >= 0:
        over = np.abs(v) > clamp
        code = np.where(over, np.uint8(2), code)
        v = np.clip(v, -clamp, clamp)
    return v, code


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, return_codes=False):
    x = np.asarray(x, dtype=np.float64)
    if b is not None:
        x = x + np.asarray(b, dtype=np.float64).reshape(1, -1, 1, 1)
    u = upfirdn2d(x, fu, up=up, padding=padding, gain=float(up * up), flip_filter=flip_filter)
    v, codes = lrelu_codes(u, gain, slope, clamp)
    y = upfirdn2d(v, fd, down=down, flip_filter=flip_filter)
    return (y, codes) if return_codes else y


def pack_codes_rowmajor(codes):
    """2-bit codes [N,C,H,W] -> uint8 [N,C,H,ceil16(W)/4], element x in byte x>>2 at bits 2*(x&3).

    Same layout as the reference's sign tensor (SG3OPS/filtered_lrelu.cpp:87-94).
    """
    n, c, h, w = codes.shape
    wp = (w + 15) & ~15
    buf = np.zeros((n, c, h, wp), dtype=np.uint8)
    buf[..., :w] = codes
    buf = buf.reshape(n, c, h, wp // 4, 4)
    return (buf[..., 0] | (buf[..., 1] << 2) | (buf[..., 2] << 4) | (buf[..., 3] << 6)).astype(np.uint8)
