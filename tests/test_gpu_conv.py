"""GPU parity tests of the MFMA convolution path (forward, data gradient, weight gradient, style and
demodulation gradients) against the CPU oracle and the reference's golden vectors.

fp32 path: the contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate), so only the
summation order differs from aten: tolerance 1e-4 x scale (north-star bar: 1e-3 max-abs).
bf16/f16 path: operands rounded to 16 bit, fp32 accumulate; compared with the fp32 oracle at 2e-2 x scale.
"""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu


def _dev(a, grad=False, dtype=None):
    t = torch.from_numpy(np.array(a)).cuda()
    if dtype is not None:
        t = t.to(dtype)
    return t.requires_grad_(True) if grad else t


def _close(a, b, tol, what=''):
    a = a.detach().float().cpu().numpy()
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1e-6, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f'{what}: max-abs err {err:.3e} (scale {scale:.3g}, tol {tol:g})'


@pytest.mark.parametrize('name', golden_names('M'))
def test_modulated_conv2d_golden(name):
    from afcm_amd.torch_utils.ops.conv2d import modulated_conv2d
    g = load_golden(name)
    demod, padding = [int(v) for v in g['meta']]
    ig = None if np.isnan(g['fmeta'][0]) else torch.tensor(float(g['fmeta'][0]), device='cuda')
    x, w, s = _dev(g['x'], True), _dev(g['w'], True), _dev(g['s'], True)
    y = modulated_conv2d(x, w, s, demodulate=bool(demod), padding=padding, input_gain=ig)
    _close(y, g['y'], 1e-4, name + ' y')
    dx, dw, ds = torch.autograd.grad((y * _dev(g['r'])).sum(), [x, w, s])
    _close(dx, g['dx'], 1e-4, name + ' dx')
    _close(dw, g['dw'], 2e-4, name + ' dw')
    _close(ds, g['ds'], 2e-4, name + ' ds')


CASES = [
    # N, I, O, k, H, W, pad
    (2, 4, 64, 3, 30, 30, 2),        # enc0-like: tiny Cin
    (2, 64, 91, 3, 22, 26, 2),       # odd Cout
    (1, 181, 128, 3, 20, 20, 2),     # odd Cin (K padding)
    (2, 96, 200, 3, 38, 38, 2),      # 38^2 plane, two o-blocks
    (2, 64, 1, 1, 32, 32, 0),        # ToRGB shape
    (2, 72, 72, 3, 36, 36, 1),       # bottleneck conv (pad 1)
    (1, 40, 48, 3, 70, 150, 2),      # wide plane: several q-chunks in the weight gradient
]


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-4), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)])
@pytest.mark.parametrize('case', CASES)
def test_scaled_conv_vs_oracle(case, dtype, tol):
    from afcm_amd.torch_utils.ops.conv2d import modulated_conv2d
    from oracle import aten_ops as ops
    n, i, o, k, h, w_, pad = case
    torch.manual_seed(hash(case) % 1000)
    x = torch.randn(n, i, h, w_)
    w = torch.randn(o, i, k, k)
    s = torch.randn(n, i) * 0.3 + 1
    if dtype != torch.float32:
        x = x.to(dtype).float()          # same 16-bit-representable inputs on both sides
    xr, wr, sr = (t.clone().requires_grad_(True) for t in (x, w, s))
    ref = ops.modulated_conv2d(xr, wr, sr, demodulate=(k == 3), padding=pad)
    r = torch.randn_like(ref)
    gref = torch.autograd.grad((ref * r).sum(), [xr, wr, sr])
    xg = x.cuda().to(dtype).requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    sg = s.cuda().requires_grad_(True)
    got = modulated_conv2d(xg, wg, sg, demodulate=(k == 3), padding=pad)
    assert got.dtype == dtype and got.shape == ref.shape
    _close(got, ref, tol, f'{case} y')
    ggot = torch.autograd.grad((got.float() * r.cuda()).sum(), [xg, wg, sg])
    for a, b, nm in zip(ggot, gref, ['dx', 'dw', 'ds']):
        _close(a, b, tol * (3 if nm != 'dx' else 1), f'{case} {nm}')


def test_plain_conv2d_gradfix():
    from afcm_amd.torch_utils.ops import conv2d_gradfix
    from oracle import aten_ops as ops
    torch.manual_seed(1)
    x = torch.randn(2, 8, 20, 20)
    w = torch.randn(12, 8, 3, 3)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = ops.conv2d(xr, wr, padding=2)
    r = torch.randn_like(ref)
    gx, gw = torch.autograd.grad((ref * r).sum(), [xr, wr])
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    got = conv2d_gradfix.conv2d(xg, wg, padding=2)
    _close(got, ref, 1e-4, 'y')
    hx, hw = torch.autograd.grad((got * r.cuda()).sum(), [xg, wg])
    _close(hx, gx, 1e-4, 'dx')
    _close(hw, gw, 2e-4, 'dw')
    with pytest.raises(NotImplementedError):
        conv2d_gradfix.conv2d(xg, wg, stride=2)


def test_full_size_layer_properties():
    """BASELINE-size modulated conv (batch 16, 512->512 @ 38^2, bf16): linearity in x and agreement of three
    random output planes with the fp32 CPU oracle computed on those planes only."""
    from afcm_amd.torch_utils.ops.conv2d import modulated_conv2d
    from oracle import aten_ops as ops
    torch.manual_seed(0)
    n, c, h = 16, 512, 36
    x = torch.randn(n, c, h, h, device='cuda', dtype=torch.bfloat16)
    w = torch.randn(c, c, 3, 3, device='cuda')
    s = torch.randn(n, c, device='cuda') * 0.2 + 1
    y = modulated_conv2d(x, w, s, padding=2)
    assert y.shape == (n, c, h + 2, h + 2)
    y2 = modulated_conv2d(x * 2, w, s, padding=2)
    assert ((y2.float() - 2 * y.float()).abs().max() <= 2e-2 * y.float().abs().max()).item()
    for (ni, oi) in [(0, 0), (7, 300), (15, 511)]:
        ref = ops.modulated_conv2d(x[ni:ni + 1].float().cpu(), w.cpu(), s[ni:ni + 1].cpu(), padding=2)
        # the batch-wide style normalisation (NET:43) uses all 16 samples: rescale the single-sample oracle
        # (it cancels under demodulation up to the 1e-8 epsilon, so plain comparison is valid)
        _close(y[ni, oi], ref[0, oi], 2e-2, f'plane {ni},{oi}')
