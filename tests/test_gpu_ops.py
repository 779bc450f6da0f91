"""GPU parity tests (run with `-m gpu` on an MI355X): HIP kernels behind the C ABI vs the CPU oracle
and vs golden vectors captured from the reference.

Tolerance: the north-star bar is <= 1e-3 max-abs in fp32; these tests assert the tighter 2e-5 x scale
(fp32 summation-order noise).  16-bit I/O is compared against the fp32 oracle with a stated looser bound.
"""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _dev(a, grad=False, dtype=None):
    if a is None:
        return None
    t = torch.from_numpy(np.array(a)).cuda()
    if dtype is not None and t.dtype.is_floating_point:
        t = t.to(dtype)
    return t.requires_grad_(True) if grad else t


def _close(a, b, tol=TOL, what=''):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f'{what}: max-abs err {err:.3e} (scale {scale:.3g}, tol {tol:g})'
    return err


def _flrelu_args(g):
    up, down, *pad = [int(v) for v in g['meta']]
    gain, slope, clamp, flip = g['fmeta']
    return dict(up=up, down=down, padding=pad, gain=float(gain), slope=float(slope),
                clamp=None if clamp < 0 else float(clamp), flip_filter=bool(flip))


@pytest.mark.parametrize('name', golden_names('F'))
def test_filtered_lrelu_golden(name):
    import warnings
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    g = load_golden(name)
    kw = _flrelu_args(g)
    x = _dev(g['x'], True)
    b = _dev(g.get('b'), True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)   # F8 (2-D filter) legitimately takes the generic GPU path
        y = flr.filtered_lrelu(x, fu=_dev(g.get('fu')), fd=_dev(g.get('fd')), b=b, **kw)
        _close(y, g['y'], what=name + ' y')
        grads = torch.autograd.grad((y * _dev(g['r'])).sum(), [x] + ([b] if b is not None else []))
    _close(grads[0], g['dx'], what=name + ' dx')
    if b is not None:
        _close(grads[1], g['db'], what=name + ' db', tol=1e-4)


@pytest.mark.parametrize('name', ['F1_up2_down2', 'F2_up2_down4', 'F3_up4_down2', 'F6_clamp', 'F10_multitile'])
def test_filtered_lrelu_sign_codes_bit_exact(name):
    """The 2-bit codes written by the kernel equal the definition-level restatement, bit for bit,
    wherever the pre-activation value is not within rounding distance of 0 or of the clamp."""
    from afcm_amd import _lib
    from afcm_amd.torch_utils.ops.filtered_lrelu import _FilteredLRelu
    from oracle import direct_np as dnp
    g = load_golden(name)
    kw = _flrelu_args(g)
    up, down, pad = kw['up'], kw['down'], kw['padding']
    x = _dev(g['x'], True)
    b = _dev(g.get('b'))
    cfg = (up, down, *pad, kw['gain'], kw['slope'], float('inf') if kw['clamp'] is None else kw['clamp'], kw['flip_filter'], 0, 0, 0)
    y = _FilteredLRelu.apply(x, _dev(g['fu']), _dev(g['fd']), b, None, cfg)
    signs = y.grad_fn.saved_tensors[2].cpu().numpy()
    xb = g['x'].astype(np.float64) + (g['b'].astype(np.float64).reshape(1, -1, 1, 1) if 'b' in g else 0)
    u = dnp.upfirdn2d(xb, g['fu'], up=up, padding=pad, gain=float(up * up), flip_filter=kw['flip_filter'])
    v, codes = dnp.lrelu_codes(u, kw['gain'], kw['slope'], kw['clamp'])
    sh, sw = signs.shape[2], signs.shape[3] * 4
    assert sh <= codes.shape[2] and codes.shape[3] <= sw
    got = np.stack([(signs >> (2 * k)) & 3 for k in range(4)], axis=-1).reshape(*signs.shape[:3], sw)
    w = codes.shape[3]
    want = codes[:, :, :sh, :]
    margin = 1e-4 * max(1.0, np.abs(u).max())
    safe = (np.abs(u[:, :, :sh]) > margin) | (u[:, :, :sh] == 0)     # exact zeros (padding) must read code 0
    if kw['clamp'] is not None:
        safe &= np.abs(np.abs(u[:, :, :sh] * kw['gain'] * np.where(u[:, :, :sh] < 0, kw['slope'], 1.0)) - kw['clamp']) > 1e-3
    assert safe.mean() > 0.95
    assert np.array_equal(got[..., :w][safe], want[safe])


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize('name', ['F1_up2_down2', 'F2_up2_down4', 'F3_up4_down2', 'F5_identity'])
def test_filtered_lrelu_16bit_io(name, dtype, tol):
    """16-bit storage, fp32 arithmetic: the error is the input/output rounding only."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    g = load_golden(name)
    kw = _flrelu_args(g)
    x16 = _dev(g['x'], dtype=dtype)
    b16 = _dev(g.get('b'), dtype=dtype)
    y = flr.filtered_lrelu(x16, fu=_dev(g.get('fu')), fd=_dev(g.get('fd')), b=b16, **kw)
    assert y.dtype == dtype
    fu = None if 'fu' not in g else torch.from_numpy(g['fu'])
    fd = None if 'fd' not in g else torch.from_numpy(g['fd'])
    ref = ops.filtered_lrelu(x16.float().cpu(), fu=fu, fd=fd, b=None if b16 is None else b16.float().cpu(), **kw)
    _close(y, ref, tol=tol, what=f'{name} {dtype}')


def test_filtered_lrelu_full_size_properties():
    """Full BASELINE size (batch 16 x 64 ch x 278^2, the enc1 shape): properties that need no CPU oracle run --
    linearity of the backward op in dy, and agreement of the fused kernel with the generic GPU path
    (upfirdn2d -> act -> upfirdn2d) on a random subset of planes checked against the CPU oracle."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    from oracle import generator as ogen
    pl = ogen.plan(256, 4, 1, {})
    L = pl['enc'][1]
    torch.manual_seed(0)
    x = torch.randn(16, 64, 278, 278, device='cuda', requires_grad=True)
    b = torch.randn(64, device='cuda') * 0.1
    kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=256.0)
    fu, fd = L['fu'].cuda(), L['fd'].cuda()
    y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=b, **kw)
    assert y.shape == (16, 64, 276, 276)
    r1, r2 = torch.randn_like(y), torch.randn_like(y)
    g1, = torch.autograd.grad(y, x, r1, retain_graph=True)
    g2, = torch.autograd.grad(y, x, r2, retain_graph=True)
    g12, = torch.autograd.grad(y, x, r1 + 2 * r2)
    assert (g12 - (g1 + 2 * g2)).abs().max().item() <= 1e-4 * g12.abs().max().item()
    # spot-check 3 planes against the CPU oracle
    for (n, c) in [(0, 0), (7, 33), (15, 63)]:
        xs = x[n:n + 1, c:c + 1].detach().cpu().requires_grad_(True)
        ref = ops.filtered_lrelu(xs, fu=L['fu'], fd=L['fd'], b=b[c:c + 1].cpu(), **kw)
        _close(y[n:n + 1, c:c + 1], ref, what=f'plane {n},{c}')
        gref, = torch.autograd.grad(ref, xs, r1[n:n + 1, c:c + 1].cpu())
        _close(g1[n:n + 1, c:c + 1], gref, what=f'plane {n},{c} dx')


def test_filtered_lrelu_errors():
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    x = torch.randn(1, 2, 8, 8, device='cuda')
    with pytest.raises(RuntimeError):
        flr.filtered_lrelu(x.cpu())                                   # no CPU path
    with pytest.raises(RuntimeError):
        flr.filtered_lrelu(x, b=torch.zeros(3, device='cuda'))        # bias length
    with pytest.raises(RuntimeError):
        flr.filtered_lrelu(x, fu=torch.ones(12, device='cuda'), up=2, padding=-20)   # upsampled buffer smaller than fd
    with pytest.raises(RuntimeError):
        flr.filtered_lrelu(x.cpu(), impl='ref')                       # 'ref' on a CPU tensor: still no CPU path


@pytest.mark.parametrize('name', ['F1_up2_down2', 'F2_up2_down4', 'F3_up4_down2', 'F4_crop', 'F6_clamp', 'F7_flip_asym', 'F8_radial2d'])
def test_impl_ref_on_gpu_tensors_runs_the_unfused_gpu_path(name):
    """Reference callers may pass impl='ref' (SG3OPS/filtered_lrelu.py:112-116); on a ROCm tensor that is the op's definition
    composed from the HIP upfirdn2d / bias_act kernels -- same golden vectors, same tolerance, y / dx / db."""
    from afcm_amd.torch_utils.ops import bias_act as ba
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from afcm_amd.torch_utils.ops import upfirdn2d as ufd
    g = load_golden(name)
    kw = _flrelu_args(g)
    x = _dev(g['x'], True)
    b = _dev(g.get('b'), True)
    y = flr.filtered_lrelu(x, fu=_dev(g.get('fu')), fd=_dev(g.get('fd')), b=b, impl='ref', **kw)
    _close(y, g['y'], what=name + ' y')
    grads = torch.autograd.grad((y * _dev(g['r'])).sum(), [x] + ([b] if b is not None else []))
    _close(grads[0], g['dx'], what=name + ' dx')
    if b is not None:
        _close(grads[1], g['db'], what=name + ' db', tol=1e-4)
    # the two other ops accept the switch as well (one device implementation each)
    xb = torch.randn(2, 3, 4, 4, device='cuda')
    assert torch.equal(ba.bias_act(xb, act='lrelu', impl='ref'), ba.bias_act(xb, act='lrelu', impl='cuda'))
    f = torch.tensor([1.0, 3.0, 3.0, 1.0], device='cuda') / 8
    assert torch.equal(ufd.upsample2d(xb, f, impl='ref'), ufd.upsample2d(xb, f, impl='cuda'))


def test_filtered_lrelu_second_order():
    """Backward is the op itself, so grad-of-grad exists; check against autograd on the oracle."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    g = load_golden('F1_up2_down2')
    kw = _flrelu_args(g)
    x = _dev(g['x'], True)
    y = flr.filtered_lrelu(x, fu=_dev(g['fu']), fd=_dev(g['fd']), b=_dev(g['b']), **kw)
    q = torch.randn(x.shape, device='cuda')
    # d/dr of <dx, q> : use r as a differentiable input
    r2 = _dev(g['r'], True)
    dx2, = torch.autograd.grad((y * r2).sum(), x, create_graph=True)
    gr, = torch.autograd.grad((dx2 * q).sum(), r2)
    xc = torch.from_numpy(g['x']).requires_grad_(True)
    rc = torch.from_numpy(g['r']).requires_grad_(True)
    yc = ops.filtered_lrelu(xc, fu=torch.from_numpy(g['fu']), fd=torch.from_numpy(g['fd']), b=torch.from_numpy(g['b']), **kw)
    dxc, = torch.autograd.grad((yc * rc).sum(), xc, create_graph=True)
    grc, = torch.autograd.grad((dxc * q.cpu()).sum(), rc)
    _close(gr, grc, what='d<dx,q>/dr')


@pytest.mark.parametrize('name', golden_names('U'))
def test_upfirdn2d_golden(name):
    from afcm_amd.torch_utils.ops import upfirdn2d as ufd
    g = load_golden(name)
    up, down, px0, px1, py0, py1, flip = [int(v) for v in g['meta']]
    gain = float(g['fmeta'][0])
    fn = str(g['fn'])
    x = _dev(g['x'], True)
    f = _dev(g['f'])
    pad = [px0, px1, py0, py1]
    if fn == 'upfirdn2d':
        y = ufd.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=bool(flip), gain=gain)
    elif fn == 'filter2d':
        y = ufd.filter2d(x, f, padding=pad, flip_filter=bool(flip), gain=gain)
    elif fn == 'upsample2d':
        y = ufd.upsample2d(x, f, up=up, padding=pad, flip_filter=bool(flip), gain=gain)
    else:
        y = ufd.downsample2d(x, f, down=down, padding=pad, flip_filter=bool(flip), gain=gain)
    _close(y, g['y'], what=name + ' y')
    dx, = torch.autograd.grad((y * _dev(g['r'])).sum(), [x])
    _close(dx, g['dx'], what=name + ' dx')


@pytest.mark.parametrize('name', golden_names('B'))
def test_bias_act_golden(name):
    from afcm_amd.torch_utils.ops import bias_act as ba
    g = load_golden(name)
    alpha, gain, clamp = [None if np.isnan(v) else float(v) for v in g['fmeta']]
    x = _dev(g['x'], True)
    b = _dev(g.get('b'), True)
    y = ba.bias_act(x, b, dim=int(g['dim']), act=str(g['act']), alpha=alpha, gain=gain, clamp=clamp)
    _close(y, g['y'], what=name)
    dx, db = torch.autograd.grad((y * _dev(g['r'])).sum(), [x, b])
    _close(dx, g['dx'], what=name + ' dx')
    _close(db, g['db'], what=name + ' db')


@pytest.mark.parametrize('act', ['lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish'])
def test_bias_act_second_order(act):
    """grad=2 mode of the kernel vs double-backward of the oracle."""
    from afcm_amd.torch_utils.ops import bias_act as ba
    from oracle import aten_ops as ops
    torch.manual_seed(3)
    xc = torch.randn(2, 3, 5, 6).requires_grad_(True)
    bc = torch.randn(3).requires_grad_(True)
    rc = torch.randn(2, 3, 5, 6)
    qc = torch.randn(2, 3, 5, 6)

    def run(x, b, r, q, fn):
        y = fn(x, b, dim=1, act=act)
        dx, = torch.autograd.grad((y * r).sum(), x, create_graph=True)
        return torch.autograd.grad((dx * q).sum(), [x, b], allow_unused=True)
    want = run(xc, bc, rc, qc, ops.bias_act)
    xg = xc.detach().cuda().requires_grad_(True)
    bg = bc.detach().cuda().requires_grad_(True)
    got = run(xg, bg, rc.cuda(), qc.cuda(), ba.bias_act)
    for a, b_, nm in zip(got, want, ['d2x', 'd2b']):
        if b_ is None:
            assert a is None or a.abs().max().item() == 0
        else:
            _close(a, b_, tol=1e-4, what=f'{act} {nm}')


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 6e-3), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize('lname', ['encoder_1', 'encoder_4', 'L10_276_128', 'L13_256_64',
                                   'encoder_11', 'encoder_12', 'L3_52_512', 'encoder_9', 'L5_84_512'])   # 52 / 54-column down-4 planes; 36^2 / 38^2 planes: the one-tile 48-row variant, forward and transposed
def test_filtered_lrelu_16bit_matrix_core_path(lname, dtype, tol):
    """16-bit activations run the matrix-core kernels (banded-Toeplitz MFMA): multi-tile planes, several (n, c) planes,
    forward and backward (sign codes in the row-quad layout) vs the fp32 CPU oracle on the same 16-bit inputs.
    Bound: 16-bit rounding of operands/intermediates only (fp32 accumulation) -- 6e-3 (f16) / 4e-2 (bf16) x scale."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    from oracle import generator as ogen
    pl = ogen.plan(256, 4, 1, {})
    L = [l for l in pl['enc'] + pl['dec'] if l['name'] == lname][0]
    h = L['in_size'] + 2
    torch.manual_seed(5)
    x = torch.randn(2, 3, h, h).to(dtype)
    b = (torch.randn(3) * 0.2).to(dtype)
    kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=256.0)
    xr = x.float().requires_grad_(True)
    ref = ops.filtered_lrelu(xr, fu=L['fu'], fd=L['fd'], b=b.float(), **kw)
    r = torch.randn_like(ref).to(dtype)
    gref, = torch.autograd.grad((ref * r.float()).sum(), xr)
    xg = x.cuda().requires_grad_(True)
    got = flr.filtered_lrelu(xg, fu=L['fu'].cuda(), fd=L['fd'].cuda(), b=b.cuda(), **kw)
    assert got.dtype == dtype and got.shape == ref.shape
    assert got.grad_fn.sign_layout == 1, 'expected the matrix-core kernel family'
    # bias gradient: summed per tile inside the backward kernel (no second pass over dx)
    bb = b.cuda().requires_grad_(True)
    got2 = flr.filtered_lrelu(xg, fu=L['fu'].cuda(), fd=L['fd'].cuda(), b=bb, **kw)
    gb, = torch.autograd.grad((got2.float() * r.cuda().float()).sum(), bb)
    bref = b.float().requires_grad_(True)
    ref2 = ops.filtered_lrelu(x.float(), fu=L['fu'], fd=L['fd'], b=bref, **kw)
    gbref, = torch.autograd.grad((ref2 * r.float()).sum(), bref)
    assert (gb.float().cpu() - gbref).abs().max().item() <= 4 * tol * max(1.0, gbref.abs().max().item()), 'fused bias gradient'
    _close(got, ref, tol=tol, what=f'{lname} {dtype} y')
    ggot, = torch.autograd.grad((got.float() * r.cuda().float()).sum(), xg)
    # leaky-ReLU kinks: 16-bit forward rounding flips the branch of elements near 0, so compare in relative L2
    d = (ggot.float().cpu() - gref)
    rel = (d.norm() / gref.norm()).item()
    assert rel <= 2 * tol, f'{lname} {dtype} dx: relative L2 {rel:.3e}'


def test_filtered_lrelu_16bit_odd_width_uses_exact_kernels():
    """Odd plane widths cannot use the aligned-pair loads of the matrix-core kernels: the op must silently take the
    other kernel family (row-major signs) and still match the oracle."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    g = load_golden('F1_up2_down2')
    kw = _flrelu_args(g)
    torch.manual_seed(2)
    x = torch.randn(1, 2, 21, 23).to(torch.bfloat16)
    ref = ops.filtered_lrelu(x.float(), fu=torch.from_numpy(g['fu']), fd=torch.from_numpy(g['fd']), **kw)
    xg = x.cuda().requires_grad_(True)
    got = flr.filtered_lrelu(xg, fu=_dev(g['fu']), fd=_dev(g['fd']), **kw)
    assert got.grad_fn.sign_layout == 0
    _close(got, ref, tol=3e-2, what='odd width bf16')
    torch.autograd.grad(got.float().sum(), xg)


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 6e-3), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize('lname', ['encoder_1', 'encoder_4', 'L10_276_128', 'encoder_11', 'encoder_12', 'L3_52_512'])
def test_filtered_lrelu_16bit_matrix_core_clamp_and_no_bias(lname, dtype, tol):
    """Matrix-core kernels, the paths the generator-shaped test does not reach: (a) no bias (the generator's convs add it
    in their epilogue), (b) inputs scaled so the clamp fires in some tiles and not in others -- the wave-uniform exact path
    of the forward kernel and the clamp-code path of the backward kernel (gradient 0 where clamped)."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    from oracle import generator as ogen
    pl = ogen.plan(256, 4, 1, {})
    L = [l for l in pl['enc'] + pl['dec'] if l['name'] == lname][0]
    h = L['in_size'] + 2
    torch.manual_seed(11)
    x = torch.randn(2, 2, h, h)
    x[0, 0] *= 40.0                      # clamp = 8 below: most of this plane clamps
    x[1, 1, : h // 2] *= 12.0            # half of this one does; the other planes never do
    x = x.to(dtype)
    kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=8.0)
    xr = x.float().requires_grad_(True)
    ref = ops.filtered_lrelu(xr, fu=L['fu'], fd=L['fd'], b=None, **kw)
    r = torch.randn_like(ref).to(dtype)
    gref, = torch.autograd.grad((ref * r.float()).sum(), xr)
    xg = x.cuda().requires_grad_(True)
    got = flr.filtered_lrelu(xg, fu=L['fu'].cuda(), fd=L['fd'].cuda(), b=None, **kw)
    assert got.grad_fn.sign_layout == 2, 'expected the wave-autonomous matrix-core kernels (no bias operand)'
    _close(got, ref, tol=tol, what=f'{lname} {dtype} clamp y')
    ggot, = torch.autograd.grad((got.float() * r.cuda().float()).sum(), xg)
    rel = ((ggot.float().cpu() - gref).norm() / gref.norm()).item()
    assert rel <= 3 * tol, f'{lname} {dtype} clamp dx: relative L2 {rel:.3e}'
    # planes that never clamp must take the fast path and still agree plane by plane
    for n, c in [(0, 1), (1, 0)]:
        d = (ggot[n, c].float().cpu() - gref[n, c]).norm() / gref[n, c].norm()
        assert d.item() <= 2 * tol, (n, c, d.item())


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-6), (torch.bfloat16, 8e-3), (torch.float16, 1e-3)])
@pytest.mark.parametrize('up,down,pad', [(1, 1, [2, 1, 2, 1]), (1, 1, [1, 2, 2, 2]), (1, 2, [1, 1, 1, 1]), (1, 2, [2, 1, 0, 3]), (2, 1, [2, 1, 2, 1]),
                                         (2, 1, [1, 2, 3, 0]), (1, 1, [-1, 3, 0, -2]), (2, 1, [-1, 4, 0, 2])])
@pytest.mark.parametrize('fshape', [(4, 4), (3, 2), (1, 4)])
def test_upfirdn2d_small_filter_tile_kernel_vs_oracle(fshape, up, down, pad, dtype, tol):
    """The LDS-tile kernel for filters of at most 4 x 4 taps (the discriminator's [1, 3, 3, 1] blur / decimation and their
    transposes) against the CPU oracle, forward and input gradient: odd plane sizes that straddle the 64 x 16 tile, uneven and
    negative paddings, asymmetric filters with and without flip."""
    from afcm_amd.torch_utils.ops import upfirdn2d as ufd
    from oracle import aten_ops as ops
    torch.manual_seed(11)
    x = torch.randn(2, 3, 37, 71)
    f = torch.randn(*fshape)
    if dtype != torch.float32:
        x = x.to(dtype).float()
    for flip in (False, True):
        xr = x.clone().requires_grad_(True)
        ref = ops.upfirdn2d(xr, f, up=up, down=down, padding=pad, flip_filter=flip, gain=up ** 2)
        r = torch.randn_like(ref)
        gref, = torch.autograd.grad((ref * r).sum(), xr)
        xg = x.cuda().to(dtype).requires_grad_(True)
        got = ufd.upfirdn2d(xg, f.cuda(), up=up, down=down, padding=pad, flip_filter=flip, gain=up ** 2)
        assert got.shape == ref.shape and got.dtype == dtype
        ref = ref.detach()
        scale = max(1.0, float(ref.abs().max()))
        assert (got.detach().float().cpu() - ref).abs().max().item() <= tol * scale, (fshape, up, down, pad, flip)
        ggot, = torch.autograd.grad((got.float() * r.cuda()).sum(), xg)
        gscale = max(1.0, float(gref.abs().max()))
        assert (ggot.float().cpu() - gref).abs().max().item() <= tol * gscale, (fshape, up, down, pad, flip, 'grad')


@pytest.mark.parametrize('dtype,tol', [(torch.bfloat16, 8e-3), (torch.float16, 1e-3)])
@pytest.mark.parametrize('up,down,pad', [(1, 1, [2, 2, 2, 2]), (1, 1, [1, 3, 2, 1]), (1, 1, [2, 3, 2, 3]), (1, 2, [1, 1, 1, 1]), (1, 2, [2, 2, 0, 3]), (2, 1, [2, 1, 2, 1]),
                                         (2, 1, [1, 2, 3, 0]), (2, 1, [2, 2, 2, 2]), (1, 1, [-1, 3, 0, -2]), (2, 1, [-1, 4, 0, 2]), (1, 2, [-2, 4, 1, 0]), (1, 1, [5, 1, 4, 0]), (2, 1, [6, 2, 7, 1]), (2, 1, [8, 2, 1, 1])])
@pytest.mark.parametrize('fshape', [(4, 4), (3, 2), (1, 4)])
@pytest.mark.parametrize('hw', [(37, 72), (9, 18), (64, 258)])
def test_upfirdn2d_small_filter_row_kernel_vs_oracle(hw, fshape, up, down, pad, dtype, tol):
    """The 16-bit row-vector kernel (csrc/upfirdn2d.hip, upfirdn2d_rows_kernel: even widths, filters of at most 4 x 4 taps, the
    discriminator's blur / decimation / their transposes) against the CPU oracle, forward and input gradient: widths of one, a few and
    many 8-column groups with a ragged last group (18, 72, 258 and whatever the padding makes of them), odd and even paddings (the odd
    ones start every 16-byte load one element early), negative paddings, row counts that leave the last strip ragged, asymmetric filters
    with and without flip.  Shapes whose output width comes out odd, and left paddings beyond what the kernel's first column group covers
    (> 3 at up 1, > 6 at up 2), take the LDS tile kernel -- the same expectations hold."""
    from afcm_amd.torch_utils.ops import upfirdn2d as ufd
    from oracle import aten_ops as ops
    torch.manual_seed(hw[1] + 7 * up + down)
    x = torch.randn(2, 3, *hw).to(dtype).float()
    f = torch.randn(*fshape)
    for flip in (False, True):
        xr = x.clone().requires_grad_(True)
        ref = ops.upfirdn2d(xr, f, up=up, down=down, padding=pad, flip_filter=flip, gain=up ** 2)
        r = torch.randn_like(ref)
        gref, = torch.autograd.grad((ref * r).sum(), xr)
        xg = x.cuda().to(dtype).requires_grad_(True)
        got = ufd.upfirdn2d(xg, f.cuda(), up=up, down=down, padding=pad, flip_filter=flip, gain=up ** 2)
        assert got.shape == ref.shape and got.dtype == dtype
        ref = ref.detach()
        scale = max(1.0, float(ref.abs().max()))
        assert (got.detach().float().cpu() - ref).abs().max().item() <= tol * scale, (fshape, up, down, pad, flip)
        ggot, = torch.autograd.grad((got.float() * r.cuda()).sum(), xg)
        gscale = max(1.0, float(gref.abs().max()))
        assert (ggot.float().cpu() - gref).abs().max().item() <= tol * gscale, (fshape, up, down, pad, flip, 'grad')


@pytest.mark.parametrize('name', ['F1_up2_down2', 'F2_up2_down4', 'F3_up4_down2', 'F4_crop', 'F6_clamp', 'F7_flip_asym'])
def test_plugin_surface_runs_the_reference_wrapper_call_sequence(name):
    """custom_ops.get_plugin (the reference's loader signature) returns the pybind-level functions on the C ABI: the forward and
    the transposed backward call exactly as SG3OPS/filtered_lrelu.py:206-217,252-263 issue them reproduce the golden y and dx."""
    from afcm_amd.torch_utils import custom_ops
    plugin = custom_ops.get_plugin(module_name='filtered_lrelu_plugin', sources=['filtered_lrelu.cpp', 'filtered_lrelu_wr.cu'],
                                   headers=['filtered_lrelu.h', 'filtered_lrelu.cu'], source_dir='.', extra_cuda_cflags=['--use_fast_math'])
    g = load_golden(name)
    kw = _flrelu_args(g)
    up, down, (px0, px1, py0, py1) = kw['up'], kw['down'], kw['padding']
    gain, slope, flip = kw['gain'], kw['slope'], kw['flip_filter']
    clamp = float('inf') if kw['clamp'] is None else kw['clamp']
    x, fu, fd, r = _dev(g['x']), _dev(g['fu']), _dev(g['fd']), _dev(g['r'])
    b = _dev(g['b']) if 'b' in g else torch.zeros(x.shape[1], device='cuda')
    empty = torch.empty([0], dtype=torch.uint8, device='cuda')
    y, so, rc = plugin.filtered_lrelu(x, fu, fd, b, empty, up, down, px0, px1, py0, py1, 0, 0, gain, slope, clamp, flip, True)
    assert rc == 0 and so.dtype == torch.uint8 and so.ndim == 4
    _close(y, g['y'], what=name + ' y (plugin)')
    fuw, fdw = fu.shape[-1], fd.shape[-1]
    fuh, fdh = (fu.shape[0], fd.shape[0]) if fu.ndim == 2 else (fuw, fdw)
    pp = [(fuw - 1) + (fdw - 1) - px0, x.shape[3] * up - y.shape[3] * down + px0 - (up - 1),
          (fuh - 1) + (fdh - 1) - py0, x.shape[2] * up - y.shape[2] * down + py0 - (up - 1)]
    sx, sy = 0 - (fuw - 1) + px0, 0 - (fuh - 1) + py0
    dx, so2, rc = plugin.filtered_lrelu(r.contiguous(), fd, fu, torch.zeros_like(b), so, down, up, *pp, sx, sy, gain * (up ** 2) / (down ** 2),
                                        slope, float('inf'), not flip, False)
    assert rc == 0 and so2.numel() == 0
    _close(dx, g['dx'], what=name + ' dx (plugin)')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('name', ['F1_up2_down2', 'F2_up2_down4', 'F3_up4_down2'])
def test_plugin_surface_reaches_the_matrix_core_kernels_for_16bit(name, dtype):
    """a maintainer who swaps only custom_ops.py gets the matrix-core kernels for 16-bit activations.  The
    shim finds the per-layer workspace itself and tags the private sign layout on ``so`` (trailing unit dimensions), which the
    reference wrapper hands back untouched for the transposed call.  y and dx vs the CPU oracle on the same 16-bit-rounded
    inputs, at 16-bit tolerance."""
    from afcm_amd.torch_utils import custom_ops
    from oracle import aten_ops as ops
    plugin = custom_ops.get_plugin(module_name='filtered_lrelu_plugin', sources=[], headers=[], source_dir='.')
    g = load_golden(name)
    kw = _flrelu_args(g)
    up, down, (px0, px1, py0, py1) = kw['up'], kw['down'], kw['padding']
    gain, slope, flip = kw['gain'], kw['slope'], kw['flip_filter']
    clamp = float('inf') if kw['clamp'] is None else kw['clamp']
    fu, fd = _dev(g['fu']), _dev(g['fd'])
    x, r = _dev(g['x']).to(dtype), _dev(g['r']).to(dtype)
    b = (_dev(g['b']) if 'b' in g else torch.zeros(x.shape[1], device='cuda')).to(dtype)
    empty = torch.empty([0], dtype=torch.uint8, device='cuda')
    y, so, rc = plugin.filtered_lrelu(x, fu, fd, b, empty, up, down, px0, px1, py0, py1, 0, 0, gain, slope, clamp, flip, True)
    assert rc == 0 and so.dtype == torch.uint8
    assert so.ndim in (5, 6), 'a 16-bit call through the plugin must run the matrix-core family (tagged sign layout)'
    xs = x.float().cpu().requires_grad_(True)
    ref = ops.filtered_lrelu(xs, fu=torch.from_numpy(g['fu']), fd=torch.from_numpy(g['fd']), b=b.float().cpu(), **kw)
    gref, = torch.autograd.grad(ref, xs, r.float().cpu())
    ref = ref.detach()
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    assert (y.float().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    fuw, fdw = fu.shape[-1], fd.shape[-1]
    fuh, fdh = (fu.shape[0], fd.shape[0]) if fu.ndim == 2 else (fuw, fdw)
    pp = [(fuw - 1) + (fdw - 1) - px0, x.shape[3] * up - y.shape[3] * down + px0 - (up - 1),
          (fuh - 1) + (fdh - 1) - py0, x.shape[2] * up - y.shape[2] * down + py0 - (up - 1)]
    sx, sy = 0 - (fuw - 1) + px0, 0 - (fuh - 1) + py0
    dx, so2, rc = plugin.filtered_lrelu(r.contiguous(), fd, fu, torch.zeros_like(b), so, down, up, *pp, sx, sy, gain * (up ** 2) / (down ** 2),
                                        slope, float('inf'), not flip, False)
    assert rc == 0 and so2.numel() == 0
    rel = ((dx.float().cpu() - gref).norm() / gref.norm()).item()         # 16-bit rounding flips branches near 0: L2, as above
    assert rel <= (8e-2 if dtype == torch.bfloat16 else 2e-2), rel
    # the unfused activation reads the reference's row-major packing only: a tagged tensor is refused, not misread
    with pytest.raises(RuntimeError):
        plugin.filtered_lrelu_act_(dx.clone(), so, 0, 0, 1.0, slope, float('inf'), False)


def test_plugin_surface_upfirdn2d_bias_act_and_error_convention():
    from afcm_amd.torch_utils import custom_ops
    from afcm_amd.torch_utils.ops.bias_act import activation_funcs
    torch.manual_seed(2)
    x = torch.randn(2, 3, 17, 19, device='cuda')
    f = torch.randn(3, 4, device='cuda')
    up = custom_ops.get_plugin('upfirdn2d_plugin', sources=[], headers=[], source_dir='.')
    y = up.upfirdn2d(x, f, 2, 2, 1, 1, 2, 1, 1, 2, False, 4.0)
    from oracle import aten_ops as ops
    ref = ops.upfirdn2d(x.cpu(), f.cpu(), up=2, down=1, padding=[2, 1, 1, 2], flip_filter=False, gain=4.0)
    assert (y.cpu() - ref).abs().max().item() <= 2e-5
    ba = custom_ops.get_plugin('bias_act_plugin', sources=[], headers=[], source_dir='.')
    b = torch.randn(3, device='cuda')
    spec = activation_funcs['lrelu']
    e = torch.empty([0], device='cuda')
    got = ba.bias_act(x, b, e, e, e, 0, 1, spec.cuda_idx, 0.2, 2 ** 0.5, 1.5)
    want = (torch.nn.functional.leaky_relu(x + b.reshape(1, -1, 1, 1), 0.2) * 2 ** 0.5).clamp(-1.5, 1.5)
    assert (got - want).abs().max().item() <= 1e-6
    fl = custom_ops.get_plugin('filtered_lrelu_plugin')
    # no fused kernel (a 40-tap filter pair): return code -1 and empty tensors, not an exception (filtered_lrelu.cpp:52-56)
    big = torch.randn(40, device='cuda')
    xs = torch.randn(1, 1, 64, 64, device='cuda')
    y, so, rc = fl.filtered_lrelu(xs, big, big, torch.zeros(1, device='cuda'), torch.empty([0], dtype=torch.uint8, device='cuda'),
                                  2, 2, 40, 39, 40, 39, 0, 0, 1.0, 0.2, float('inf'), False, True)
    assert rc == -1 and y.numel() == 0 and so.numel() == 0
    with pytest.raises(RuntimeError):
        custom_ops.get_plugin('conv2d_gradfix_plugin')


@pytest.mark.parametrize('name', ['F1_up2_down2', 'F3_up4_down2', 'F6_clamp'])
def test_registered_custom_ops_filtered_lrelu(name):
    """torch.ops.afcm.filtered_lrelu (dispatcher-registered, plugin argument list of filtered_lrelu.cpp:16-18): forward with sign
    write and the transposed backward with sign read reproduce the golden y and dx."""
    import afcm_amd  # noqa: F401  (registers the operator library)
    g = load_golden(name)
    kw = _flrelu_args(g)
    up, down, (px0, px1, py0, py1) = kw['up'], kw['down'], kw['padding']
    gain, slope, flip = kw['gain'], kw['slope'], kw['flip_filter']
    clamp = float('inf') if kw['clamp'] is None else kw['clamp']
    x, fu, fd, r = _dev(g['x']), _dev(g['fu']), _dev(g['fd']), _dev(g['r'])
    b = _dev(g['b']) if 'b' in g else torch.zeros(x.shape[1], device='cuda')
    empty = torch.empty([0], dtype=torch.uint8, device='cuda')
    y, so, rc = torch.ops.afcm.filtered_lrelu(x, fu, fd, b, empty, up, down, px0, px1, py0, py1, 0, 0, gain, slope, clamp, flip, True)
    assert rc == 0
    _close(y, g['y'], what=name + ' y (torch.ops.afcm)')
    fuw, fdw = fu.shape[-1], fd.shape[-1]
    pp = [(fuw - 1) + (fdw - 1) - px0, x.shape[3] * up - y.shape[3] * down + px0 - (up - 1),
          (fuw - 1) + (fdw - 1) - py0, x.shape[2] * up - y.shape[2] * down + py0 - (up - 1)]
    dx, _, rc = torch.ops.afcm.filtered_lrelu(r.contiguous(), fd, fu, torch.zeros_like(b), so, down, up, *pp, px0 - (fuw - 1), py0 - (fuw - 1),
                                              gain * (up ** 2) / (down ** 2), slope, float('inf'), not flip, False)
    assert rc == 0
    _close(dx, g['dx'], what=name + ' dx (torch.ops.afcm)')


def test_registered_custom_ops_upfirdn2d_bias_act_conv():
    """torch.ops.afcm.{upfirdn2d, bias_act, conv2d_pack_weights, conv2d, conv2d_wgrad} against the CPU oracle / aten; CPU tensors
    are refused by the dispatcher (no CPU kernel is registered)."""
    import afcm_amd  # noqa: F401
    from afcm_amd.torch_utils.ops.bias_act import activation_funcs
    from oracle import aten_ops as ops
    torch.manual_seed(4)
    x = torch.randn(2, 3, 17, 19, device='cuda')
    f = torch.randn(3, 4, device='cuda')
    y = torch.ops.afcm.upfirdn2d(x, f, 2, 2, 1, 1, 2, 1, 1, 2, False, 4.0)
    ref = ops.upfirdn2d(x.cpu(), f.cpu(), up=2, down=1, padding=[2, 1, 1, 2], flip_filter=False, gain=4.0)
    assert (y.cpu() - ref).abs().max().item() <= 2e-5
    b = torch.randn(3, device='cuda')
    e = torch.empty([0], device='cuda')
    got = torch.ops.afcm.bias_act(x, b, e, e, e, 0, 1, activation_funcs['lrelu'].cuda_idx, 0.2, 2 ** 0.5, 1.5)
    want = (torch.nn.functional.leaky_relu(x + b.reshape(1, -1, 1, 1), 0.2) * 2 ** 0.5).clamp(-1.5, 1.5)
    assert (got - want).abs().max().item() <= 1e-6
    # in-place activation with sign write (filtered_lrelu.cpp:213): x is mutated, the schema declares it
    xa = torch.randn(1, 2, 8, 20, device='cuda')
    want_a = (torch.nn.functional.leaky_relu(xa, 0.2) * 1.5).clamp(-1.0, 1.0)
    so = torch.ops.afcm.filtered_lrelu_act_(xa, torch.empty([0], dtype=torch.uint8, device='cuda'), 0, 0, 1.5, 0.2, 1.0, True)
    assert so.dtype == torch.uint8 and (xa - want_a).abs().max().item() <= 1e-6
    # convolution triple, fp32 (exact MFMA path)
    xc = torch.randn(2, 5, 12, 14, device='cuda')
    w = torch.randn(7, 5, 3, 3, device='cuda')
    osc = torch.rand(2, 7, device='cuda') + 0.5
    ob = torch.randn(7, device='cuda')
    wp, rows_pad = torch.ops.afcm.conv2d_pack_weights(w, torch.float32, 0)
    yc = torch.ops.afcm.conv2d(xc, wp, osc, ob, 7, 3, 2, rows_pad)
    refc = torch.nn.functional.conv2d(xc.cpu().double(), w.cpu().double(), padding=2) * osc.cpu().double()[:, :, None, None] + ob.cpu().double()[None, :, None, None]
    _close(yc, refc.float(), tol=1e-5, what='afcm::conv2d')
    dy = torch.randn_like(yc)
    wpt, rows_pad_t = torch.ops.afcm.conv2d_pack_weights(w, torch.float32, 1)
    dxc = torch.ops.afcm.conv2d(dy, wpt, None, None, 5, 3, 0, rows_pad_t)
    xr = xc.cpu().double().requires_grad_(True)
    wr = w.cpu().double().requires_grad_(True)
    torch.nn.functional.conv2d(xr, wr, padding=2).backward(dy.cpu().double())
    _close(dxc, xr.grad.float(), tol=1e-5, what='afcm::conv2d (data gradient)')
    dw = torch.ops.afcm.conv2d_wgrad(dy, xc, 7, 5, 3, 2)
    _close(dw, wr.grad.float(), tol=1e-5, what='afcm::conv2d_wgrad')
    with pytest.raises(NotImplementedError):
        torch.ops.afcm.upfirdn2d(x.cpu(), f.cpu(), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize('shape', [(2, 5, 8, 12), (3, 7, 4, 4), (2, 5, 7, 9), (4, 24)])
@pytest.mark.parametrize('act', ['lrelu', 'swish', 'linear'])
def test_bias_act_vector_and_element_paths(act, shape, dtype, tol):
    """The 16-byte kernel (plane size a multiple of 8 / 4 elements: one bias per vector) and the element kernel (ragged
    planes) against the CPU oracle on the same rounded inputs: value, first- and second-order gradients, several bias channels."""
    from afcm_amd.torch_utils.ops import bias_act as ba
    from oracle import aten_ops as ops
    torch.manual_seed(9)
    dim = 1
    xc = torch.randn(*shape).to(dtype).float().requires_grad_(True)
    bc = torch.randn(shape[dim]).to(dtype).float().requires_grad_(True)
    rc, qc = torch.randn(*shape).to(dtype).float(), torch.randn(*shape).to(dtype).float()
    # clamp only in fp32: the backward masks on the SAVED output, and a 16-bit output that rounds onto the clamp value flips the mask
    kw = dict(dim=dim, act=act, gain=1.3, clamp=(2.0 if (act != 'linear' and dtype == torch.float32) else None))

    def run(x, b, r, q, fn):
        y = fn(x, b, **kw)
        dx, = torch.autograd.grad((y.float() * r.float()).sum(), x, create_graph=True)
        d2 = torch.autograd.grad((dx.float() * q.float()).sum(), [x, b], allow_unused=True) if dx.requires_grad else (None, None)
        return y, dx, d2
    yw, dxw, d2w = run(xc, bc, rc, qc, ops.bias_act)
    xg = xc.detach().to(dtype).cuda().requires_grad_(True)
    bg = bc.detach().to(dtype).cuda().requires_grad_(True)
    yg, dxg, d2g = run(xg, bg, rc.to(dtype).cuda(), qc.to(dtype).cuda(), ba.bias_act)
    _close(yg, yw, tol=tol, what=f'{act} {shape} y')
    _close(dxg, dxw, tol=tol, what=f'{act} {shape} dx')
    for a, b_, nm in zip(d2g, d2w, ['d2x', 'd2b']):
        if b_ is None:
            assert a is None or a.abs().max().item() == 0
        elif a is not None:
            _close(a, b_, tol=8 * tol, what=f'{act} {shape} {nm}')


@pytest.mark.parametrize('n,kw,kg,couts', [(16, 512, 1024, [512, 362, 181, 91, 64, 1]), (5, 32, 1024, [8, 8, 7]), (20, 64, 0, [33, 16])])
def test_affine_bank_matches_the_layers_one_by_one(n, kw, kg, couts):
    """torch_utils/ops/affine_bank.py (C ABI afcm_affine_bank_*): the styles of several SynthesisLayers' affine FCs (NET:349-352,
    FullyConnectedLayer NET:69-104, ToRGB factor NET:351) from one launch and every gradient from three, against the FullyConnectedLayer
    modules called one by one on cat(w, global_w): forward 1e-5, gradients 1e-4 relative (fp32 sums in a different order); a batch of 20
    goes through the 16-row kernels twice (second pass accumulates the weight gradients); layers without a gradient contribute zeros."""
    from afcm_amd.networks_stylegan3 import FullyConnectedLayer
    from afcm_amd.torch_utils.ops import affine_bank as ab
    torch.manual_seed(3)
    nl = len(couts)
    fcs = [FullyConnectedLayer(kw + kg, c, bias_init=1).cuda() for c in couts]
    for fc in fcs:
        with torch.no_grad():
            fc.bias.add_(torch.randn_like(fc.bias) * 0.1)
    ws = torch.randn(n, nl + 2, kw, device='cuda', requires_grad=True)
    g = torch.randn(n, kg, device='cuda', requires_grad=True) if kg else None
    scales = [1.0] * (nl - 1) + [0.125]
    specs = [ab.Spec(fc, 1 + l, sc) for l, (fc, sc) in enumerate(zip(fcs, scales))]
    assert ab.supported(ws, g, specs)
    got = ab.affine_bank(ws, g, specs)
    want = [fc(torch.cat((ws[:, 1 + l], g), 1) if kg else ws[:, 1 + l]) * sc for l, (fc, sc) in enumerate(zip(fcs, scales))]
    for a, b in zip(got, want):
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
    rs = [torch.randn_like(b) for b in want]
    rs[1] = None                                      # this layer's styles receive no gradient
    params = [p for fc in fcs for p in (fc.weight, fc.bias)]
    ins = [ws] + ([g] if kg else []) + params
    loss_g = sum((a * r).sum() for a, r in zip(got, rs) if r is not None)
    loss_w = sum((b * r).sum() for b, r in zip(want, rs) if r is not None)
    gg = torch.autograd.grad(loss_g, ins, allow_unused=True)
    gw = torch.autograd.grad(loss_w, ins, allow_unused=True)
    for i, (a, b) in enumerate(zip(gg, gw)):
        if b is None:
            assert a is None or a.abs().max().item() == 0.0, i
            continue
        assert a is not None, i
        assert (a - b).abs().max().item() <= 1e-4 * max(1e-3, b.abs().max().item()), (i, (a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize('n', [1, 16])
def test_modulation_bank_is_bit_identical_to_the_layers_one_by_one(n):
    """torch_utils/ops/modulation_bank.py (C ABI afcm_modulation_bank_*): (w_hat, in_scale, out_scale) of a list of layers (NET:41-57,
    346-352) from 2 launches and the weight / style gradients from 3, against ``modulation_coefficients_fused`` layer by layer: the same
    kernel bodies, so every output and gradient must match bit for bit.  Covers the three row-length classes of the weight kernels
    (<= 1024, <= 4608, longer), a 1x1 layer, a layer without magnitude, the layer that does not demodulate (ToRGB) and output gradients
    that are missing (None) or present for w_hat / in_scale / out_scale independently."""
    from afcm_amd.torch_utils.ops import modulation_bank as mb
    from afcm_amd.torch_utils.ops.conv2d import modulation_coefficients_fused
    torch.manual_seed(11)
    shapes = [(64, 32, 3, True), (37, 100, 3, True), (16, 512, 3, True), (24, 1024, 3, True), (8, 700, 1, True), (3, 64, 1, False), (512, 512, 3, True)]
    ws = [torch.randn(o, i, k, k, device='cuda', requires_grad=True) for o, i, k, _ in shapes]
    ts = [torch.randn(n, i, device='cuda', requires_grad=True) for _, i, _, _ in shapes]
    mags = [None if l == 1 else torch.rand([], device='cuda') + 0.5 for l in range(len(shapes))]
    items = [mb.Item(w, t, m, dm) for w, t, m, (_, _, _, dm) in zip(ws, ts, mags, shapes)]
    assert mb.supported(items)
    got = mb.modulation_bank(items)
    want = [modulation_coefficients_fused(w, t, demodulate=dm, magnitude=m) for w, t, m, (_, _, _, dm) in zip(ws, ts, mags, shapes)]
    for l, (a, b) in enumerate(zip(got, want)):
        for x, y in zip(a, b):
            assert (x is None) == (y is None), l
            if x is not None:
                assert torch.equal(x, y), (l, (x - y).abs().max().item())
    # gradients: a random cotangent on every output, except layer 2 (none on w_hat), layer 3 (none on out_scale), layer 4 (in_scale only)
    def loss(mods):
        tot = 0
        gen = torch.Generator(device='cuda').manual_seed(5)
        for l, (w_hat, s, d) in enumerate(mods):
            rw = torch.randn(w_hat.shape, device='cuda', generator=gen)
            rs = torch.randn(s.shape, device='cuda', generator=gen)
            rd = torch.randn(d.shape, device='cuda', generator=gen) if d is not None else None
            if l not in (2, 4):
                tot = tot + (w_hat * rw).sum()
            tot = tot + (s * rs).sum()
            if d is not None and l not in (3, 4):
                tot = tot + (d * rd).sum()
        return tot
    gg = torch.autograd.grad(loss(got), ws + ts, allow_unused=True)
    gw = torch.autograd.grad(loss(want), ws + ts, allow_unused=True)
    for i, (a, b) in enumerate(zip(gg, gw)):
        assert (a is None) == (b is None), i
        if a is not None:
            assert torch.equal(a, b), (i, (a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize('layer,h,w,flip', [('enc1', 201, 113, False), ('enc1', 97, 50, True), ('enc4', 230, 262, False), ('enc4', 75, 118, True),
                                           ('dec3', 61, 58, False), ('dec3', 120, 27, True)])
def test_fp32_strip_kernel_on_ragged_planes(layer, h, w, flip):
    """csrc/filtered_lrelu.hip flrelu_strip_kernel (fp32: one wave marches down a column strip) on plane shapes that are no multiple of
    anything: several strips with a ragged last one (48 / 56 / 104 output columns per strip), two or three row segments (96 rows each),
    odd widths and heights, flipped filters -- forward, sign-reading backward and bias gradient against the CPU oracle at 2e-5 of the
    output's scale (the golden fixtures F1-F12 are small: one strip, one segment).  The three generator configurations: up 2 / down 2,
    up 2 / down 4 (two column blocks per lane), up 4 / down 2."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    from oracle import generator as ogen
    pl = ogen.plan(256, 4, 1, {})
    L = {'enc1': pl['enc'][1], 'enc4': pl['enc'][4], 'dec3': pl['dec'][3]}[layer]
    torch.manual_seed(h * 1000 + w)
    x = torch.randn(1, 3, h, w) * 2.0
    b = torch.randn(3) * 0.3
    kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=3.0, flip_filter=flip)
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = ops.filtered_lrelu(xr, fu=L['fu'], fd=L['fd'], b=br, **kw)
    xg, bg = x.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    got = flr.filtered_lrelu(xg, fu=L['fu'].cuda(), fd=L['fd'].cuda(), b=bg, **kw)
    _close(got, ref, what=f'{layer} {h}x{w} y')
    r = torch.randn_like(ref)
    gref = torch.autograd.grad((ref * r).sum(), [xr, br])
    ggot = torch.autograd.grad((got * r.cuda()).sum(), [xg, bg])
    _close(ggot[0], gref[0], what=f'{layer} {h}x{w} dx')
    _close(ggot[1], gref[1], tol=1e-4, what=f'{layer} {h}x{w} db')


@pytest.mark.parametrize('layer', ['enc1', 'enc4', 'dec3'])
def test_fp32_strip_kernel_hands_a_nan_on(layer):
    """a NaN activation must stay a NaN (the reference kernel and the aten path propagate it, filtered_lrelu.cu:484-572); the
    fast activation of the fp32 strip kernel clamped through v_med3_f32, which returns -clamp for a NaN operand.  One NaN input sample:
    every output the oracle makes NaN is NaN here, nothing beyond one output pixel around them (the strip kernel's polyphase tables carry
    one zero tap, and NaN * 0 = NaN: its footprint is 12 x 12 where the zero-skipping definition gives 11 x 11), equal values elsewhere."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    from oracle import generator as ogen
    pl = ogen.plan(256, 4, 1, {})
    L = {'enc1': pl['enc'][1], 'enc4': pl['enc'][4], 'dec3': pl['dec'][3]}[layer]
    torch.manual_seed(7)
    x = torch.randn(1, 2, 70, 66)
    x[0, 1, 33, 29] = float('nan')
    kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=256.0)
    ref = ops.filtered_lrelu(x, fu=L['fu'], fd=L['fd'], b=None, **kw)
    got = flr.filtered_lrelu(x.cuda(), fu=L['fu'].cuda(), fd=L['fd'].cuda(), b=None, **kw).cpu()
    assert ref.isnan().any() and not ref[0, 0].isnan().any()
    gn, rn = got.isnan(), ref.isnan()
    assert not (rn & ~gn).any(), f'{layer}: {int((rn & ~gn).sum())} outputs are NaN in the oracle and finite here'
    near = torch.nn.functional.max_pool2d(rn.float(), 3, stride=1, padding=1) > 0
    assert not (gn & ~near).any(), f'{layer}: {int(gn.sum())} NaN outputs, the oracle has {int(rn.sum())}'
    both = ~gn & ~rn
    _close(torch.where(both, got, torch.zeros_like(got)), torch.where(both, ref, torch.zeros_like(ref)), what=f'{layer} y beside the NaN')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(16, 512, 36, 36), (2, 3, 8, 12), (1, 5, 4, 4)], ids=str)
def test_pool_blocks_matches_adaptive_avg_pool(shape, dtype):
    """networks_stylegan3._PoolBlocks (C ABI afcm_pool_blocks_fwd / _bwd): AdaptiveAvgPool2d((4, 4)) of planes that divide evenly (NET:636,683),
    fp32 block means straight from the 16-bit activations, and its gradient, against aten on the same (rounded) values."""
    from afcm_amd.networks_stylegan3 import _PoolBlocks
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(shape, generator=g).to(dtype)
    r = torch.randn([shape[0], shape[1], 4, 4], generator=g)
    xr = x.float().requires_grad_(True)
    ref = torch.nn.functional.adaptive_avg_pool2d(xr, (4, 4))
    gref, = torch.autograd.grad((ref * r).sum(), [xr])
    xg = x.cuda().requires_grad_(True)
    got = _PoolBlocks.apply(xg)
    ggot, = torch.autograd.grad((got * r.cuda()).sum(), [xg])
    assert got.dtype == torch.float32 and ggot.dtype == dtype
    assert float((got.cpu() - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))
    tol = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 1e-6}[dtype]
    assert float((ggot.float().cpu() - gref).abs().max()) <= tol * float(gref.abs().max())
