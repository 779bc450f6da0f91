"""Row-pitched activation tensors (afcm_amd/torch_utils/ops/_rows.py, C ABI ``*_pitch`` arguments): every kernel that takes a pitch
gives bit-identical results on a pitched and on a dense copy of the same data, whatever the padding holds (NaN-filled here), and
the kernels that write a pitched tensor fill its padding with finite values (include/afcm_hip.h, afcm_conv2d_wgrad_ld)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pitched(t, fill=float('nan')):
    """A row-pitched copy of t whose padding columns hold ``fill``."""
    from afcm_amd.torch_utils.ops import _rows
    n, c, h, w = t.shape
    ld = (w + 31) // 32 * 32 + 32                 # some pitch, whatever the layout rule would choose for this width
    buf = torch.full([n, c, h, ld], fill, dtype=t.dtype, device=t.device)
    buf[..., :w] = t
    v = buf[..., :w]
    assert _rows.pitch_of(v) == ld and not v.is_contiguous()
    return v


def _plan():
    from oracle import generator as ogen
    return ogen.plan(256, 4, 1, {})


def _padding_of(v):
    from afcm_amd.torch_utils.ops import _rows
    return _rows.whole_buffer(v)[..., v.shape[3]:]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('layer,h,ch', [('enc1', 278, 5), ('enc4', 150, 3), ('enc12', 38, 7), ('dec3', 38, 4), ('dec10', 86, 3)])
def test_filtered_lrelu_pitched_equals_dense(layer, h, ch, dtype, monkeypatch):
    """Forward (sign write, skip + per-plane factor epilogue) and the transposed backward (sign read, plane sums) of the wave kernels
    on pitched x / skip / y vs dense tensors: bit-identical y, signs, dx, plane sums; y's padding finite."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from afcm_amd.torch_utils.ops import _rows, fused_layer
    monkeypatch.setattr(_rows, 'MAX_OVERHEAD', 10.0)       # pitched outputs for the narrow planes too (the layout rule leaves them dense)
    pl = _plan()
    L = (pl['enc'] if layer.startswith('enc') else pl['dec'])[int(layer[3:])]
    torch.manual_seed(3)
    x = torch.randn(2, ch, h, h, device='cuda').to(dtype)
    fu, fd = L['fu'].cuda(), L['fd'].cuda()
    cfg = fused_layer._cfg(L['up'], L['down'], L['padding'], math.sqrt(2), 0.2, 256.0)
    osc = torch.rand(2, ch, device='cuda') + 0.5
    y0, s0, lay0, _ = flr._run(x, fu, fd, None, None, cfg, True, oscale=osc)
    skip = torch.randn_like(y0)
    y0, s0, lay0, _ = flr._run(x, fu, fd, None, None, cfg, True, oscale=osc, skip=skip)
    assert lay0 == 2 and y0.is_contiguous()
    y1, s1, lay1, _ = flr._run(_pitched(x), fu, fd, None, None, cfg, True, oscale=osc, skip=_pitched(skip), pitched_out=True)
    assert lay1 == 2 and not y1.is_contiguous() and y1.stride(2) % 32 == 0
    assert torch.equal(y0, y1) and torch.equal(s0, s1)
    assert torch.isfinite(_padding_of(y1).float()).all()
    # the transposed op, as the fused node's backward issues it
    g = torch.randn_like(y0)
    bcfg = flr._backward_cfg(cfg, fu, fd, x.shape, y0.shape, 2)
    d0, _, _, p0 = flr._run(g, fd, fu, None, s0, bcfg, False, want_plane_sum=True, oscale=osc)
    d1, _, _, p1 = flr._run(_pitched(g), fd, fu, None, s0, bcfg, False, want_plane_sum=True, oscale=osc, pitched_out=True)
    assert d0.shape == x.shape and not d1.is_contiguous()
    assert torch.equal(d0, d1) and torch.equal(p0, p1)
    assert torch.isfinite(_padding_of(d1).float()).all()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
# (276 -> 278-wide outputs on 288-element rows: the tile width of conv2d_fwd16x_kernel's scalar-granule epilogue, r06;
#  70 / 64 / 130 / 91 output channels: the 128-, 64-, 128 + 64- and 96-row blocks)
@pytest.mark.parametrize('cin,cout,h', [(20, 70, 150), (64, 64, 38), (33, 130, 86), (8, 64, 276), (16, 91, 276), (12, 130, 276)])
def test_conv_and_wgrad_pitched_equal_dense(cin, cout, h, dtype, monkeypatch):
    """3x3 pad-2 conv (forward kernel = data-gradient kernel) and the weight gradient on pitched operands with NaN padding vs dense."""
    from afcm_amd.torch_utils.ops import _rows
    from afcm_amd.torch_utils.ops import conv2d as conv
    monkeypatch.setattr(_rows, 'MAX_OVERHEAD', 10.0)
    torch.manual_seed(5)
    n = 2
    x = torch.randn(n, cin, h, h, device='cuda').to(dtype)
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.1
    osc = torch.rand(n, cout, device='cuda') + 0.5
    bias = torch.randn(cout, device='cuda')
    wp, rows_pad = conv.pack_weights(w, dtype, 0)
    y0 = conv._conv_raw(x, wp, rows_pad, osc, cout, 3, 2, obias=bias)
    y1 = conv._conv_raw(_pitched(x), wp, rows_pad, osc, cout, 3, 2, obias=bias, pitched_out=True)
    assert y0.is_contiguous() and not y1.is_contiguous()
    assert torch.equal(y0, y1)
    y2 = conv._conv_raw(_pitched(x), wp, rows_pad, osc, cout, 3, 2, obias=bias)           # pitched in, dense out
    assert y2.is_contiguous() and torch.equal(y0, y2)
    dy = torch.randn_like(y0)
    # dy's padding: finite up to the next multiple of 8 columns (what the producing kernels guarantee), NaN beyond
    dyp = _pitched(dy)
    q = dy.shape[3]
    _rows.whole_buffer(dyp)[..., q:(q + 7) // 8 * 8] = 3e4           # (finite in float16 too)
    dw0 = conv._wgrad_raw(dy, x, cout, cin, 3, 2)
    dw1 = conv._wgrad_raw(dyp, _pitched(x), cout, cin, 3, 2)
    assert torch.isfinite(dw1).all()
    assert torch.equal(dw0, dw1)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('h,w', [(278, 278), (36, 36), (5, 22)])
def test_plane_dot_pitched(h, w, dtype):
    from afcm_amd.torch_utils.ops import conv2d as conv
    from afcm_amd.torch_utils.ops import _rows
    torch.manual_seed(7)
    a = torch.randn(3, 5, h, w, device='cuda').to(dtype)
    b = torch.randn(3, 5, h, w, device='cuda').to(dtype)
    want = (a.double() * b.double()).sum(dim=(2, 3))
    ld = (w + 63) // 64 * 64 + 64
    def pitched(t):
        buf = torch.full([3, 5, h, ld], float('nan'), dtype=dtype, device='cuda')
        buf[..., :w] = t
        return buf[..., :w]
    for (pa, pb) in [(pitched(a), pitched(b)), (pitched(a), b), (a, pitched(b))]:
        got = conv.plane_dot(pa, pb)
        assert torch.isfinite(got).all()
        assert (got.double() - want).abs().max().item() <= 1e-4 * want.abs().max().item() + 1e-3
    got = conv.plane_dot(pitched(a))
    assert (got.double() - a.double().sum(dim=(2, 3))).abs().max().item() <= 1e-3 * math.sqrt(h * w)
    sp = conv.scale_planes(pitched(a), torch.rand(3, 5, device='cuda') + 0.5)
    assert _rows.pitch_of(sp) == ld and torch.isfinite(sp.float()).all()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_plane_dot_operands_outside_the_pitched_kernels_preconditions(dtype):
    """an expanded (stride-0) operand, an odd-width 16-bit plane behind a pitch, and a dense operand at an odd element
    offset next to a pitched one must fall through to the dense kernel (on a copy), not raise from afcm_plane_dot_ld."""
    from afcm_amd.torch_utils.ops import conv2d as conv
    torch.manual_seed(11)
    for (h, w) in [(9, 21), (12, 24)]:
        a = torch.randn(2, 3, h, w, device='cuda').to(dtype)
        buf = torch.full([2, 3, h, 64], float('nan'), dtype=dtype, device='cuda')
        buf[..., :w] = a
        pa = buf[..., :w]
        g = torch.randn(2, 3, 1, 1, device='cuda').to(dtype).expand(2, 3, h, w)             # stride-0 gradient
        flat = torch.randn(2 * 3 * h * w + 1, device='cuda').to(dtype)
        odd = flat[1:].view(2, 3, h, w)                                                      # contiguous, odd element offset
        for x, y in [(pa, g), (g, pa), (pa, odd), (odd, pa), (g, None)]:
            want = (x.double() * (y.double() if y is not None else 1.0)).sum(dim=(2, 3))
            got = conv.plane_dot(x, y)
            assert (got.double() - want).abs().max().item() <= 1e-4 * want.abs().max().item() + 1e-3, (h, w, dtype)


def test_fused_layer_pitched_equals_dense(monkeypatch):
    """The fused layer node with the row-pitched layout on and off: same z, same gradients (bit-identical: the kernels do the same
    arithmetic in the same order, only the addresses differ)."""
    from afcm_amd.torch_utils.ops import _rows
    from afcm_amd.torch_utils.ops import fused_layer
    pl = _plan()
    L = pl['enc'][5]
    torch.manual_seed(11)
    n, cin, cout, h = 2, 24, 40, L['in_size']
    x0 = torch.randn(n, cin, h, h, device='cuda').to(torch.bfloat16)
    w0 = torch.randn(cout, cin, 3, 3, device='cuda') * 0.1
    ins = torch.rand(n, cin, device='cuda') + 0.5
    outs = torch.rand(n, cout, device='cuda') + 0.5
    nxt = torch.rand(n, cout, device='cuda') + 0.5
    b0 = torch.randn(cout, device='cuda') * 0.1
    fu, fd = L['fu'].cuda(), L['fd'].cuda()

    def run(enabled):
        monkeypatch.setattr(_rows, 'ENABLED', enabled)
        x, w, i, o, nx, b = (t.clone().requires_grad_(True) for t in (x0, w0, ins, outs, nxt, b0))
        z = fused_layer.conv_filtered_lrelu(x, w, i, o, b, fu, fd, L['up'], L['down'], L['padding'], math.sqrt(2), 0.2, 256.0, 2, next_scale=nx)
        assert z.is_contiguous() != enabled
        skipz = torch.randn(z.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(1)).to(z.dtype)
        # a second node consumes z (prescaled by nx) and an encoder-style skip of its own
        L2 = pl['enc'][6]
        w2 = (torch.randn(48, cout, 3, 3, device='cuda', generator=torch.Generator('cuda').manual_seed(2)) * 0.1).requires_grad_(True)
        z2 = fused_layer.conv_filtered_lrelu(z, w2, nx, None, None, L2['fu'].cuda(), L2['fd'].cuda(), L2['up'], L2['down'], L2['padding'], math.sqrt(2), 0.2, 256.0, 2,
                                             prescaled=True)
        r = torch.randn(z2.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(3)).to(z2.dtype)
        grads = torch.autograd.grad([(z2.float() * r.float()).sum() + (z.float() * skipz.float()).sum()], [x, w, i, o, nx, b, w2])
        return [z.detach().contiguous(), z2.detach().contiguous()] + [g_.contiguous() for g_ in grads]
    dense, pitched = run(False), run(True)
    names = ['z', 'z2', 'dx', 'dw', 'd_in', 'd_out', 'd_next', 'db', 'dw2']
    for nm, a, b in zip(names, dense, pitched):
        if nm in ('z', 'z2', 'dx', 'dw', 'dw2'):
            assert torch.equal(a, b), nm
        else:                                       # plane dot products: another summation order
            assert (a.float() - b.float()).abs().max().item() <= 2e-2 * a.float().abs().max().item(), nm
