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
    (2, 64, 96, 3, 20, 34, 1),       # pad 1, non-square
    (1, 40, 48, 3, 70, 150, 2),      # wide plane: several q-chunks in the weight gradient
    # output widths around the weight gradient's 16-pixel K groups and 64-pixel chunks (dead-group skipping, read-ahead across
    # groups) and the forward kernel's 8-pixel store granules (transposing epilogue, ragged right edge)
    (1, 64, 64, 3, 6, 14, 2),        # Q = 16: one live group per wave pair
    (1, 64, 64, 3, 6, 30, 2),        # Q = 32
    (1, 64, 64, 3, 6, 46, 2),        # Q = 48: three live groups
    (1, 64, 64, 3, 6, 62, 2),        # Q = 64: exactly one chunk
    (1, 64, 64, 3, 6, 64, 2),        # Q = 66: second chunk with 2 live pixels
    (2, 64, 128, 3, 6, 98, 2),       # Q = 100: 36 live pixels in the second chunk, two o-tiles
    # 65 .. 96 output rows take the 96-row block of the 16x16x32 kernel (forward: Cout, data gradient: Cin)
    (2, 91, 64, 3, 22, 26, 2),       # data gradient with 91 rows
    (1, 64, 80, 3, 40, 54, 2),       # 80 rows: the last 16-row fragment of each wave half empty
    (2, 96, 65, 3, 12, 150, 2),      # 65 rows forward, 96 rows in the data gradient, 50-wide tiles
    # weight gradient: tiles whose last 32 rows / columns lie outside the matrix form a class of their own (fewer, longer K shares;
    # their idle wave quadrants are skipped) -- a partial row of tiles, a partial column, both with the corner tile, one live quadrant only
    (2, 91, 91, 3, 20, 22, 2),       # 2 x 2 tiles: one full, a partial column, a partial row, the corner
    (1, 130, 100, 3, 14, 30, 2),     # I = 130: third tile column with 2 live columns; O = 100: 36 live rows (full class)
    (2, 160, 20, 3, 10, 70, 2),      # O = 20: every tile partial (one class again), I = 160: 32 live columns in the last
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


@pytest.mark.parametrize('dtype,tol', [(torch.bfloat16, 4e-2), (torch.float16, 8e-3)])
@pytest.mark.parametrize('case', ['decoder_skip_next', 'decoder_prescaled', 'encoder'])
def test_fused_layer_node_matches_op_by_op(case, dtype, tol):
    """conv(+bias) -> filtered_lrelu(+skip, x next styles) as ONE autograd node (torch_utils/ops/fused_layer.py) vs the
    reference composition of the same layer run in fp32 on the CPU oracle (NET:366-377): output and every gradient
    (x, w, styles, demodulation, bias, skip, next styles)."""
    from afcm_amd.torch_utils.ops import fused_layer
    from oracle import aten_ops as ops
    from oracle import generator as ogen
    pl = ogen.plan(256, 4, 1, {})
    L = [l for l in pl['dec'] if l['name'] == 'L9_148_181'][0] if case != 'encoder' else pl['enc'][5]
    torch.manual_seed(7)
    n, cin, cout, h = 2, 24, 16, L['in_size']
    x = torch.randn(n, cin, h, h)
    w = torch.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)
    s = torch.rand(n, cin) + 0.5
    d = torch.rand(n, cout) + 0.5
    b = torch.randn(cout) * 0.2
    oh = L['out_size']
    skip = torch.randn(n, cout, oh, oh) if case == 'decoder_skip_next' else None
    ns = (torch.rand(n, cout) + 0.5) if case == 'decoder_skip_next' else None
    modulated = case != 'encoder'
    prescaled = case == 'decoder_prescaled'
    act = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=256.0)

    # fp32 reference on the 16-bit-rounded inputs
    def q(t):
        return None if t is None else t.to(dtype).float()
    leaves = {k: (v.clone().requires_grad_(True) if v is not None else None)
              for k, v in dict(x=q(x), w=w, s=s if modulated else None, d=d if modulated else None, b=b, skip=q(skip), ns=ns).items()}
    xs = leaves['x'] * leaves['s'][:, :, None, None] if modulated else leaves['x']
    y = torch.nn.functional.conv2d(xs, leaves['w'], padding=2)
    if modulated:
        y = y * leaves['d'][:, :, None, None]
    z = ops.filtered_lrelu(y, fu=L['fu'], fd=L['fd'], b=leaves['b'], **act)
    if skip is not None:
        z = (z + leaves['skip']) * leaves['ns'][:, :, None, None]
    r = torch.randn_like(z).to(dtype).float()
    names = [k for k, v in leaves.items() if v is not None]
    gref = dict(zip(names, torch.autograd.grad((z * r).sum(), [leaves[k] for k in names])))

    dev = {k: (v.detach().cuda().requires_grad_(True) if v is not None else None)
           for k, v in dict(x=x.to(dtype), w=w, s=s if modulated else None, d=d if modulated else None, b=b,
                            skip=None if skip is None else skip.to(dtype), ns=ns).items()}
    xin = dev['x']
    if prescaled:      # the producer applied the styles: feed s * x and expect the gradient w.r.t. that product
        xin = (x.to(dtype).float() * s[:, :, None, None]).to(dtype).cuda().requires_grad_(True)
    assert fused_layer.available(xin, dev['w'], L['fu'].cuda(), L['fd'].cuda(), conv_pad=2, **act)
    got = fused_layer.conv_filtered_lrelu(xin, dev['w'], dev['s'], dev['d'], dev['b'], L['fu'].cuda(), L['fd'].cuda(), conv_pad=2,
                                          skip=dev['skip'], next_scale=dev['ns'], prescaled=prescaled, **act)
    assert got.dtype == dtype
    _close_rel(got, z, tol, f'{case} z')
    wanted = [k for k in names if not (prescaled and k in ('x', 's'))]
    ggot = dict(zip(wanted, torch.autograd.grad((got.float() * r.cuda()).sum(), [dev[k] for k in wanted])))
    for k in wanted:
        _close_rel(ggot[k], gref[k], 3 * tol, f'{case} d{k}')
    if prescaled:
        gx, = torch.autograd.grad((fused_layer.conv_filtered_lrelu(xin, dev['w'], dev['s'], dev['d'], dev['b'], L['fu'].cuda(), L['fd'].cuda(),
                                                                   conv_pad=2, prescaled=True, **act).float() * r.cuda()).sum(), [xin])
        # d/d(s*x) = (d/dx) / s
        _close_rel(gx, gref['x'] / s[:, :, None, None], 3 * tol, 'prescaled dxs')


@pytest.mark.parametrize('dtype,tol', [(torch.bfloat16, 4e-2), (torch.float16, 8e-3)])
def test_demodulation_gradient_by_homogeneity_falls_back_on_clamped_planes(dtype, tol):
    """The fused node's <dys, y> (the demodulation gradient, NET:50-52 backward) comes from <g, z> - s_next <g, skip> where no strip of
    a plane could reach the clamp (filtered_lrelu is then positively homogeneous of degree 1) and from the real dot product on the
    others, decided on the device from the flags the sign-writing kernels emit (C ABI afcm_filtered_lrelu_args.clamp_flags,
    afcm_plane_dot_gated_ld).  Half of the planes are scaled far below the clamp, the other half far into it: the flags say so, the
    flagged planes' gradient is bit-identical to the all-real-dot-products path, the others agree to 16-bit rounding, and both match
    the fp32 oracle (which clamps, filtered_lrelu.cu:484-572)."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from afcm_amd.torch_utils.ops import fused_layer
    from oracle import aten_ops as ops
    from oracle import generator as ogen
    pl = ogen.plan(256, 4, 1, {})
    L = [l for l in pl['dec'] if l['name'] == 'L9_148_181'][0]
    torch.manual_seed(11)
    n, cin, cout, h = 2, 16, 8, L['in_size']
    x = torch.randn(n, cin, h, h)
    w = torch.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)
    s = torch.rand(n, cin) + 0.5
    d = torch.where(torch.arange(n * cout).reshape(n, cout) % 2 == 0, torch.full([n, cout], 0.05), torch.full([n, cout], 40.0)) * (torch.rand(n, cout) + 0.5)
    b = torch.randn(cout) * 0.01
    oh = L['out_size']
    skip = torch.randn(n, cout, oh, oh)
    ns = torch.rand(n, cout) + 0.5
    act = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=4.0)
    xq, skq = x.to(dtype).float(), skip.to(dtype).float()
    lv = [t.clone().requires_grad_(True) for t in (xq, w, s, d, b, skq, ns)]
    y = torch.nn.functional.conv2d(lv[0] * lv[2][:, :, None, None], lv[1], padding=2) * lv[3][:, :, None, None]
    z = (ops.filtered_lrelu(y, fu=L['fu'], fd=L['fd'], b=lv[4], **act) + lv[5]) * lv[6][:, :, None, None]
    r = torch.randn_like(z).to(dtype).float()
    gd_ref, gns_ref = torch.autograd.grad((z * r).sum(), [lv[3], lv[6]])

    def run(homog):
        fused_layer.HOMOGENEOUS_DOT = homog
        try:
            dev = [t.detach().cuda().requires_grad_(True) for t in (x.to(dtype), w, s, d, b, skip.to(dtype), ns)]
            got = fused_layer.conv_filtered_lrelu(dev[0], dev[1], dev[2], dev[3], dev[4], L['fu'].cuda(), L['fd'].cuda(), conv_pad=2,
                                                  skip=dev[5], next_scale=dev[6], **act)
            flags = got.grad_fn.saved_tensors[11]
            gd, gns = torch.autograd.grad((got.float() * r.cuda()).sum(), [dev[3], dev[6]])
            return got, flags, gd.cpu(), gns.cpu()
        finally:
            fused_layer.HOMOGENEOUS_DOT = True
    got, flags, gd_h, gns_h = run(True)
    _, flags0, gd_r, gns_r = run(False)
    assert flags0 is None and flags is not None and flags.dtype == torch.int32 and tuple(flags.shape[:2]) == (n, cout)
    flagged = (flags.sum(dim=2) > 0).cpu()
    small = (torch.arange(n * cout).reshape(n, cout) % 2 == 0)
    assert flagged[~small].all(), 'every strongly driven plane must be flagged'
    assert not flagged[small].any(), 'planes far below the clamp must not be flagged'
    assert torch.equal(gd_h[flagged], gd_r[flagged]), 'flagged planes take the real dot product'
    assert torch.equal(gns_h, gns_r)
    _close_rel(gd_h[~flagged], gd_r[~flagged], tol, 'homogeneity vs real dot products, planes below the clamp')
    _close_rel(gd_h, gd_ref, 3 * tol, 'dd vs the oracle')
    _close_rel(gns_h, gns_ref, 3 * tol, 'd(next styles) vs the oracle')


def _close_rel(a, b, tol, what):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    rel = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
    assert rel <= tol, f'{what}: relative L2 error {rel:.3e} (tol {tol:g})'


@pytest.mark.parametrize('dims', [(37, 45, 5), (181, 200, 4), (8, 600, 2), (40, 512, 16)])   # several 32-o / 64-i chunks with ragged last ones; rows beyond / exactly at the register-resident weight-norm limit (512 x 9)
@pytest.mark.parametrize('demod', [True, False])
def test_fused_modulation_coefficients_match_eager(demod, dims):
    """afcm_weight_norm_* / afcm_style_coefs_* (one launch each way) vs the eager torch restatement of NET:41-57:
    values and gradients w.r.t. the weights and the raw styles, for both outputs."""
    from afcm_amd.torch_utils.ops.conv2d import modulation_coefficients, modulation_coefficients_fused
    torch.manual_seed(3)
    o, i, n = dims
    w = torch.randn(o, i, 3, 3, device='cuda')
    t = torch.randn(n, i, device='cuda') + 1.0
    mag = torch.tensor(1.7, device='cuda')
    rs, rd, rw = torch.randn(n, i, device='cuda'), torch.randn(n, o, device='cuda'), torch.randn(o, i, 3, 3, device='cuda')

    def run(fn, **kw):
        wl, tl = w.clone().requires_grad_(True), t.clone().requires_grad_(True)
        w_hat, s_eff, d = fn(wl, tl, demodulate=demod, **kw)
        loss = (s_eff * rs).sum() + (w_hat * rw).sum()
        if d is not None:
            loss = loss + (d * rd).sum()
        gw, gt = torch.autograd.grad(loss, [wl, tl])
        return w_hat, s_eff, d, gw, gt
    want = run(modulation_coefficients, input_gain=mag.rsqrt())
    got = run(modulation_coefficients_fused, magnitude=mag)
    for a, b, nm in zip(got, want, ['w_hat', 's_eff', 'd', 'dw', 'dt']):
        if b is None:
            assert a is None
            continue
        _close(a, b, 2e-5, f'demod={demod} {nm}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('hw', [(7, 7), (1, 3), (38, 38), (278, 278), (5, 64), (86, 86), (150, 150)])
def test_plane_dot_matches_torch(dtype, hw):
    """afcm_plane_dot: per-plane <a, b> and plane sums; odd plane sizes put every plane on a different 16-byte phase."""
    from afcm_amd.torch_utils.ops.conv2d import plane_dot
    torch.manual_seed(5)
    a = torch.randn(3, 5, *hw, device='cuda').to(dtype)
    b = torch.randn(3, 5, *hw, device='cuda').to(dtype)
    want = (a.double() * b.double()).sum([2, 3])
    got = plane_dot(a, b)
    assert got.dtype == torch.float32 and got.shape == (3, 5)
    scale = (a.double().abs() * b.double().abs()).sum([2, 3])
    assert ((got.double() - want).abs() <= 1e-5 * scale + 1e-6).all()
    got1 = plane_dot(a)
    assert ((got1.double() - a.double().sum([2, 3])).abs() <= 1e-5 * a.double().abs().sum([2, 3]) + 1e-6).all()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cout,cin,ks', [(7, 5, 3), (91, 128, 3), (64, 4, 3), (1, 64, 1), (181, 70, 3)])
def test_pack_weights_layout(dtype, cout, cin, ks):
    """afcm_conv2d_pack_weights / _pack_weights2: [ceil(cols / BK)][k*k][rows_pad][BK], zero padded; mode 0 rows = cout, mode 1
    rows = cin with the taps flipped (include/afcm_hip.h) -- both images of the one-launch form equal the single-mode calls and
    a definition-level restatement, bit for bit."""
    from afcm_amd import _lib
    from afcm_amd.torch_utils.ops.conv2d import pack_weights, pack_weights_both
    torch.manual_seed(3)
    w = torch.randn(cout, cin, ks, ks, device='cuda')
    bk = _lib.load().afcm_conv2d_block_k_ks(_lib._DTYPES[dtype], ks)

    def restate(mode):
        src = w if mode == 0 else w.transpose(0, 1).flip([2, 3])
        rows, cols = src.shape[:2]
        rows_pad, nkc = (rows + 63) // 64 * 64, (cols + bk - 1) // bk
        full = torch.zeros(rows_pad, nkc * bk, ks * ks, device='cuda')
        full[:rows, :cols] = src.reshape(rows, cols, ks * ks)
        return full.reshape(rows_pad, nkc, bk, ks * ks).permute(1, 3, 0, 2).contiguous().to(dtype), rows_pad

    (p0, r0), (p1, r1) = pack_weights_both(w, dtype)
    for mode, (got, rp) in enumerate([(p0, r0), (p1, r1)]):
        want, rows_pad = restate(mode)
        single, rp_single = pack_weights(w, dtype, mode)
        assert rp == rows_pad == rp_single and got.shape == want.shape == single.shape
        assert torch.equal(got, want), f'mode {mode}: one-launch image differs from the layout definition'
        assert torch.equal(single, want), f'mode {mode}: single-mode image differs from the layout definition'


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('ks', [3, 1])
def test_pack_weights_bank_is_bit_identical_to_the_layers_one_by_one(dtype, ks):
    """conv2d.pack_weights_bank (C ABI afcm_conv2d_pack_bank): the forward and data-gradient MFMA images of a list of weights from ONE
    launch against pack_weights_both layer by layer -- same kernel body behind a layer index, so every byte must match.  Shapes: channel
    counts that are no multiple of the 16-channel K chunk or of the 64-row padding, a 1-output layer, the 512 x 512 bottleneck."""
    from afcm_amd.torch_utils.ops import conv2d as C
    torch.manual_seed(2)
    shapes = [(64, 4), (91, 64), (128, 91), (181, 128), (512, 512), (1, 64), (37, 100)]
    ws = [torch.randn(o, i, ks, ks, device='cuda') for o, i in shapes]
    got = C.pack_weights_bank(ws, dtype)
    for w, ((d0, rp0), (d1, rp1)) in zip(ws, got):
        (e0, q0), (e1, q1) = C.pack_weights_both(w, dtype)
        assert (rp0, rp1) == (q0, q1)
        # rows past cout / cin inside the 64-row padding are never read by the conv kernels' valid outputs, but both routes zero them
        assert torch.equal(d0.view(torch.uint8), e0.view(torch.uint8)) and torch.equal(d1.view(torch.uint8), e1.view(torch.uint8)), tuple(w.shape)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,cin,cout,h,w,pad', [(2, 32, 64, 34, 34, 0), (1, 37, 130, 67, 130, 0), (2, 16, 16, 20, 36, 1), (1, 64, 128, 130, 130, 0),
                                               (3, 8, 200, 9, 12, 2)])
def test_stride2_conv_equals_the_decimated_stride1_result(dtype, n, cin, cout, h, w, pad):
    """conv2d.strided_conv2d (C ABI afcm_conv2d_stride2, csrc/conv2d.hip conv2d_fwd16s2_kernel: the discriminator's down-sampling convs,
    CoModGAN/generator.py:613-692).  r05: the kernel is held to the ORACLE at layer level -- oracle/discriminator.py conv2d_resample's
    strided branch (conv2d_resample.py:130-134: F.conv2d(x, w, stride=down)) in float64 on the SAME 16-bit-rounded operands: fp32
    accumulation + one output rounding: at most half a unit in the last place of the largest binade, 2^-8 (bfloat16) / 2^-11 (float16) of the output's scale -- and to the stride-1 MFMA route it
    replaced (the stride-1 result sliced [::2, ::2]; since r05 a different kernel with a different K order: the same bound, no longer bit
    for bit).  Input and weight gradients bit for bit against that route (both run the same stride-1 kernels on the same zero-stuffed
    dy), and the double backward an R1 penalty takes (gradient of |dy/dx|^2 w.r.t. the weights) to 1e-3 of its scale.
    Shapes: channel tails, several 128-row blocks, odd heights, tiles at all four image edges, paddings 0 / 1 / 2."""
    from afcm_amd.torch_utils.ops import conv2d as C
    torch.manual_seed(7)
    x = torch.randn(n, cin, h, w, device='cuda').to(dtype).requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)).requires_grad_(True)
    assert C.strided_conv2d_supported(x, wt, pad)
    got = C.strided_conv2d(x, wt, pad)
    want = C.scaled_conv2d(x, wt, None, None, pad)[:, :, ::2, ::2]
    # the oracle's strided branch on the operands the kernel multiplies (x as given, the weights rounded to the activation dtype)
    ref = torch.nn.functional.conv2d(x.detach().double().cpu(), wt.detach().to(dtype).double().cpu(), stride=2, padding=pad)
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    scale = ref.abs().max().item()
    for name, y in (('stride-2 kernel', got), ('stride-1 route', want)):
        assert y.shape == ref.shape, name
        err = (y.detach().double().cpu() - ref).abs().max().item()
        assert err <= 1.05 * ulp * scale, (name, err, scale)
    r = torch.randn_like(got)
    gg = torch.autograd.grad((got.float() * r.float()).sum(), [x, wt], create_graph=True)
    gw = torch.autograd.grad((want.float() * r.float()).sum(), [x, wt], create_graph=True)
    assert torch.equal(gg[0], gw[0]) and torch.equal(gg[1], gw[1])
    # second order: d/dw of |dL/dx|^2 (the R1 pattern)
    hg, = torch.autograd.grad(gg[0].float().square().sum(), [wt])
    hw, = torch.autograd.grad(gw[0].float().square().sum(), [wt])
    assert (hg - hw).abs().max().item() <= 1e-3 * max(1e-6, hw.abs().max().item())


def _close16(a, b):
    """Two 16-bit results of the same fp32-accumulated sums in different orders: within 1.5 ulp (bfloat16) of the tensor's scale -- far
    below what a stale weight image produces (the weights differ by tens of percent in the tests that use this)."""
    a, b = a.float(), b.float()
    return (a - b).abs().max().item() <= 1.5 * 2.0 ** -8 * max(1e-12, b.abs().max().item())


def test_stride2_conv_repacks_after_an_optimizer_step():
    """The stride-2 conv keeps its packed weight image while the weight tensor is unchanged (keyed on storage and version counter);
    the fused Adam kernel writes parameters through raw pointers and therefore bumps the version counters itself (afcm_amd/optim.py):
    after a step the conv must run on the NEW weights, and an in-place torch update must be seen as well."""
    from afcm_amd.optim import FusedScrubAdam
    from afcm_amd.torch_utils.ops import conv2d as C
    torch.manual_seed(3)
    x = torch.randn(2, 16, 20, 20, device='cuda').to(torch.bfloat16)
    wt = torch.nn.Parameter(torch.randn(24, 16, 3, 3, device='cuda') / 12)
    opt = FusedScrubAdam([wt], lr=0.05, betas=(0.0, 0.99))
    y0 = C.strided_conv2d(x, wt, 1)
    assert torch.equal(y0, C.strided_conv2d(x, wt, 1))                 # (second call: the cached image)
    v0 = wt._version
    y0.float().square().sum().backward()
    opt.step()
    assert wt._version > v0, 'the optimizer kernel must bump the version counter of what it wrote'
    y1 = C.strided_conv2d(x, wt, 1)
    want = C.scaled_conv2d(x, wt.detach(), None, None, 1)[:, :, ::2, ::2]
    assert _close16(y1, want) and not _close16(y1, y0)
    with torch.no_grad():
        wt.mul_(0.5)
    assert _close16(C.strided_conv2d(x, wt, 1), C.scaled_conv2d(x, wt.detach(), None, None, 1)[:, :, ::2, ::2])


def test_stride2_conv_under_address_reuse_uses_current_weights():
    """every real caller hands the stride-2 conv a TEMPORARY (`self.weight * self.weight_gain`, then `.to(float32)`:
    networks_discriminator.py Conv2dLayer -> conv2d_resample), whose address the caching allocator hands out again -- to the next
    iteration's temporary after an optimizer step, or to another same-shape layer's.  A cache keyed on the address served a stale image.
    Here: two same-shape layers under no_grad, temporaries freed in between so that addresses repeat, then a weight update between
    iterations; every result against the stride-1 route on the CURRENT weights."""
    from afcm_amd.torch_utils.ops import conv2d as C
    from afcm_amd.torch_utils.ops import conv2d_resample as R
    torch.manual_seed(11)
    x = torch.randn(2, 16, 20, 20, device='cuda').to(torch.bfloat16)
    wa = torch.nn.Parameter(torch.randn(24, 16, 3, 3, device='cuda') / 12)
    wb = torch.nn.Parameter(torch.randn(24, 16, 3, 3, device='cuda') / 12)
    gain = 0.37

    def layer(w):                                          # what Conv2dLayer.forward does (down = 2 without a filter)
        return R._conv2d_wrapper(x, w * gain, stride=2, padding=1)

    def want(w):
        return C.scaled_conv2d(x, (w.detach() * gain).float(), None, None, 1)[:, :, ::2, ::2]

    seen = set()
    with torch.no_grad():
        for it in range(6):
            for w in (wa, wb):
                t = w * gain
                seen.add(t.data_ptr())
                del t
                got = layer(w)
                assert _close16(got, want(w)), (it, 'stale packed image')
            wa.mul_(1.25)                                  # "optimizer step": the next iteration's temporaries reuse the addresses
            wb.add_(0.01)
    assert len(seen) < 12, 'the allocator never reused an address: the test did not exercise the hazard'


# ---- fp32 on the 16-bit matrix pipe: split operands (C ABI afcm_split16 / afcm_conv2d_split) -------------------------------------
def test_split16_parts_sum_back_to_the_fp32_value():
    from afcm_amd.torch_utils.ops import conv2d as C
    g = torch.Generator().manual_seed(3)
    x = (torch.randn([2, 5, 6, 10], generator=g) * torch.logspace(-20, 20, 600).view(2, 5, 6, 10)).cuda()
    x[0, 0, 0, :4] = torch.tensor([float('inf'), -float('inf'), float('nan'), 0.0])
    sc = torch.rand([2, 5], generator=g).cuda() + 0.5
    for scale in (None, sc):
        v = x if scale is None else x * scale[:, :, None, None]
        fin = torch.isfinite(v)
        for k in (2, 3):
            parts = C.split16(x, scale, k, torch.bfloat16)
            assert parts.shape == (k, 2, 5, 6, 10) and parts.dtype == torch.bfloat16
            tot = parts.double().sum(0)
            # round-to-nearest parts: k parts leave at most 2^-(8 k + 1) of the value (three parts: 2^-25 -- below half an fp32 ulp)
            err = ((tot - v.double()).abs() / v.double().abs().clamp_min(1e-300))[fin]
            assert float(err.max()) <= 2.0 ** -(8 * k + 1) * 1.01, (k, float(err.max()))
            assert torch.equal(parts[0][~fin].float().isnan(), v[~fin].isnan()) and torch.equal(parts[0][~fin].float().isinf(), v[~fin].isinf())
            assert float(parts[1:][:, ~fin].float().abs().max()) == 0.0       # inf / nan stay in the leading part only
    # float16 parts under the power-of-two factor of the magnitude bound: 22 bits where the second part is a normal number
    # (|gs v| >= 2^-3), an absolute 2^-25 (in scaled units) below -- 2^-40 of the bound
    y = (torch.randn([2, 5, 6, 10], generator=g) * torch.logspace(-12, 3, 600).view(2, 5, 6, 10)).cuda()
    for scale in (None, sc):
        v = (y if scale is None else y * scale[:, :, None, None]).double()
        word = C.amax_bits(y, scale)
        gs = C.pow2_factor(word)
        bound = float((y if scale is None else y * scale[:, :, None, None]).abs().max())
        assert word.view(torch.float32).item() == bound and 2.0 ** 14 <= bound * gs < 2.0 ** 15 and float(np.frexp(gs)[0]) == 0.5      # a power of two
        parts = C.split16(y, scale, 2, torch.float16, word)
        assert parts.dtype == torch.float16 and bool(torch.isfinite(parts.float()).all())
        err = (parts.double().sum(0) - v * gs).abs()
        big = (v * gs).abs() >= 2.0 ** -3
        assert float((err[big] / (v * gs).abs()[big]).max()) <= 2.0 ** -22 and float(err[~big].max()) <= 2.0 ** -25


def test_amax_bits_edge_cases():
    from afcm_amd.torch_utils.ops import conv2d as C
    z = torch.zeros([1, 2, 4, 6]).cuda()
    assert C.amax_bits(z).item() == 0 and np.isfinite(C.pow2_factor(C.amax_bits(z)))      # all zeros: some finite power of two
    for bad in (float('inf'), float('nan')):
        t = torch.randn([1, 2, 4, 6]).cuda()
        t[0, 1, 2, 3] = bad
        fin = t[torch.isfinite(t)]
        # r06: a non-finite element does not count towards the bound (one inf used to take the factor away from every finite element)
        assert C.amax_bits(t).view(torch.float32).item() == float(fin.abs().max()) and C.pow2_factor(C.amax_bits(t)) > 1.0
        parts = C.split16(t, None, 2, torch.float16, C.amax_bits(t))
        assert torch.equal(parts[0].float().isnan(), t.isnan()) and torch.equal(parts[0].float().isinf(), t.isinf())
    t = torch.randn([3, 7, 5, 9]).cuda()                                                     # 945 elements: the scalar path and an unaligned view
    for v in (t, t.flatten()[1:].view(1, 1, 8, 118)):
        v = v.contiguous()
        assert C.amax_bits(v).view(torch.float32).item() == float(v.abs().max())
    big = torch.randn([4, 8, 64, 66]).cuda()                                                 # vector path, per-plane factors, a word raised twice
    sc = (torch.rand([4, 8]).cuda() - 0.5) * 8
    w1 = C.amax_bits(big, sc)
    assert w1.view(torch.float32).item() == float((big * sc[:, :, None, None]).abs().max())
    w2 = C.amax_bits(big * 3, None, out=C.amax_bits(big))
    assert w2.view(torch.float32).item() == float((big * 3).abs().max())
    tiny = torch.full([1, 1, 2, 2], 1e-38).cuda()
    assert np.isfinite(C.pow2_factor(C.amax_bits(tiny)))


_SPLIT_MODES = [((torch.float16, 3, 3, 3), 1.5e-6), ((torch.bfloat16, 6, 6, 6), 1.5e-6), ((torch.bfloat16, 3, 3, 3), 2e-5)]


@pytest.mark.parametrize('mode,tol', _SPLIT_MODES, ids=['f16x3', 'bf16x6', 'bf16x3'])
@pytest.mark.parametrize('case', [(2, 4, 64, 30, 30, 2), (2, 64, 91, 22, 26, 2), (1, 181, 128, 20, 20, 2), (2, 72, 72, 36, 36, 1),
                                  (1, 40, 48, 70, 150, 2), (1, 512, 512, 36, 36, 2)], ids=str)
def test_fp32_conv_on_split_16bit_operands(case, mode, tol):
    """The fp32 3x3 conv, its data gradient and its weight gradient through the split-operand route against float64 aten on the
    CPU.  Scaled float16 x 3 terms and bfloat16 x 6 terms: as good as an fp32 dot product (tolerance = fp32 accumulation over
    K = 9 Cin); bfloat16 x 3 terms: ~16 bits."""
    from afcm_amd.torch_utils.ops import conv2d as C
    n, i, o, h, w, pad = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn([n, i, h, w], generator=g) * torch.exp(1.5 * torch.randn([n, i, 1, 1], generator=g))      # planes of very different size
    wt = torch.randn([o, i, 3, 3], generator=g) / (3 * i ** 0.5)
    si, so = torch.rand([n, i], generator=g) + 0.5, torch.rand([n, o], generator=g) + 0.5
    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    sid, sod = si.double().requires_grad_(True), so.double().requires_grad_(True)
    yd = torch.nn.functional.conv2d(xd * sid[:, :, None, None], wd, padding=pad) * sod[:, :, None, None]
    r = torch.randn(yd.shape, generator=g) * 1e-3                                                               # small gradients
    ref = torch.autograd.grad((yd * r.double()).sum(), [xd, wd, sid, sod])
    old = C.FP32_SPLIT
    C.FP32_SPLIT = mode
    try:
        xg, wg = x.cuda().requires_grad_(True), wt.cuda().requires_grad_(True)
        sig, sog = si.cuda().requires_grad_(True), so.cuda().requires_grad_(True)
        y = C._ScaledConv2d.apply(xg, wg, sig, sog, pad, False)
        got = torch.autograd.grad((y * r.cuda()).sum(), [xg, wg, sig, sog])
    finally:
        C.FP32_SPLIT = old
    assert y.dtype == torch.float32
    _close(y, yd.float(), tol, 'y')
    for nm, a, b, f in zip(('dx', 'dw', 'd in_scale', 'd out_scale'), got, ref, (1, 3, 3, 3)):
        _close(a, b.float(), f * tol, nm)


def test_fp32_split_route_is_the_default_and_native_kernels_remain():
    from afcm_amd.torch_utils.ops import conv2d as C
    assert C.FP32_SPLIT is not None
    g = torch.Generator().manual_seed(5)
    x, wt = torch.randn([1, 32, 18, 20], generator=g).cuda(), (torch.randn([48, 32, 3, 3], generator=g) / 17).cuda()
    y_split = C._ScaledConv2d.apply(x, wt, None, None, 2, False)
    old, C.FP32_SPLIT = C.FP32_SPLIT, None
    try:
        y_native = C._ScaledConv2d.apply(x, wt, None, None, 2, False)
    finally:
        C.FP32_SPLIT = old
    ref = torch.nn.functional.conv2d(x.double().cpu(), wt.double().cpu(), padding=2).float()
    _close(y_native, ref, 2e-6, 'native fp32 MFMA')
    _close(y_split, ref, 2e-6, 'split operands')
    assert not torch.equal(y_split, y_native)          # different kernels, different summation order
    # odd widths keep the native kernel
    xo = torch.randn([1, 32, 18, 21], generator=g).cuda()
    _close(C._ScaledConv2d.apply(xo, wt, None, None, 2, False), torch.nn.functional.conv2d(xo.double().cpu(), wt.double().cpu(), padding=2).float(), 2e-6, 'odd width')
    # an all-zero and a non-finite input go through (zeros; NaN where the window touches the NaN, nothing else)
    z = C._ScaledConv2d.apply(torch.zeros_like(x), wt, None, None, 2, False)
    assert float(z.abs().max()) == 0.0
    xn = x.clone()
    xn[0, 3, 9, 10] = float('nan')
    yn = C._ScaledConv2d.apply(xn, wt, None, None, 2, False)
    assert bool(yn[0, :, 9:12, 10:13].isnan().all())
    # r06: one inf does not take the split's power-of-two factor away from the finite elements -- outside the inf's
    # window the result is as good as without it (with g = 1 elements below 6e-5 lost their second part: errors of 1e-4 of scale here)
    xs = x * 1e-3
    xi = xs.clone()
    xi[0, 3, 9, 10] = float('inf')
    yi = C._ScaledConv2d.apply(xi, wt, None, None, 2, False)
    refs = torch.nn.functional.conv2d(xs.double().cpu(), wt.double().cpu(), padding=2).float()
    keep = torch.ones_like(refs, dtype=torch.bool)
    keep[0, :, 9:12, 10:13] = False
    assert not bool(torch.isfinite(yi[0, :, 9:12, 10:13]).any()) and bool(torch.isfinite(yi.cpu()[keep]).all())
    assert float((yi.cpu() - refs)[keep].abs().max()) <= 2e-6 * float(refs.abs().max())


@pytest.mark.parametrize('dtype,k', [(torch.float16, 2), (torch.bfloat16, 3)])
@pytest.mark.parametrize('shape', [(2, 5, 6, 10), (1, 3, 7, 9), (2, 4, 38, 38)], ids=str)
def test_plane_dot_of_split_parts(dtype, k, shape):
    """afcm_plane_dot_parts: <x, b> per plane from the 16-bit parts of scale * x, factor undone (vector and scalar paths)."""
    from afcm_amd.torch_utils.ops import conv2d as C
    g = torch.Generator().manual_seed(13)
    x, b = torch.randn(shape, generator=g).cuda() * 37.0, torch.randn(shape, generator=g).cuda()
    sc = (torch.rand(shape[:2], generator=g) + 0.5).cuda()
    if shape[3] % 2:                                     # split16 itself takes any width; the conv route needs even ones
        pass
    bound = C.amax_bits(x, sc) if dtype == torch.float16 else None
    parts = C.split16(x, sc, k, dtype, bound)
    got = C.plane_dot_parts(parts, b, bound)
    want = ((x.double() * sc.double()[:, :, None, None]) * b.double()).sum([2, 3])
    size = ((x.double() * sc.double()[:, :, None, None]).abs() * b.double().abs()).sum([2, 3])
    assert float(((got.double() - want).abs() / size).max()) <= (2e-6 if dtype == torch.float16 or k == 3 else 2e-4)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_demodulation_gradient_takes_the_real_dot_product_when_the_skip_branch_dwarfs_the_layer(dtype):
    """<dys, y> = d (<g, z> - s_next <g, skip>) is a difference of two nearly equal plane sums when |skip| >> |F(y)|, each with
    the 16-bit rounding of z, so the error of d_out grows by |skip| / |F|.  The gate (afcm_plane_dot_gated_ld) sends a plane whose sums
    cancel below 1/8 of their size to the real dot product: with a skip 200 x the layer's output every plane must come out bit-identical
    to the all-real-dot-products path (and no plane is flagged by the clamp); with a skip of the layer's own size the homogeneous route is
    still taken (results differ in rounding only)."""
    from afcm_amd.torch_utils.ops import fused_layer
    from oracle import generator as ogen
    pl = ogen.plan(256, 4, 1, {})
    L = [l for l in pl['dec'] if l['name'] == 'L9_148_181'][0]
    torch.manual_seed(13)
    n, cin, cout, h = 2, 16, 8, L['in_size']
    x = torch.randn(n, cin, h, h).to(dtype)
    w = torch.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)
    s = torch.rand(n, cin) + 0.5
    d = (torch.rand(n, cout) + 0.5) * 0.05                      # far below the clamp: no plane is flagged
    b = torch.randn(cout) * 0.01
    oh = L['out_size']
    ns = torch.rand(n, cout) + 0.5
    act = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=4.0)
    r = torch.randn(n, cout, oh, oh).to(dtype).float().cuda()

    def run(homog, skip):
        fused_layer.HOMOGENEOUS_DOT = homog
        try:
            dev = [t.detach().cuda().requires_grad_(True) for t in (x, w, s, d, b, skip.to(dtype), ns)]
            got = fused_layer.conv_filtered_lrelu(dev[0], dev[1], dev[2], dev[3], dev[4], L['fu'].cuda(), L['fd'].cuda(), conv_pad=2,
                                                  skip=dev[5], next_scale=dev[6], **act)
            flags = got.grad_fn.saved_tensors[11]
            (gd,) = torch.autograd.grad((got.float() * r).sum(), [dev[3]])
            return flags, gd.cpu()
        finally:
            fused_layer.HOMOGENEOUS_DOT = True
    base = torch.randn(n, cout, oh, oh)
    big = base * 10.0                                          # |F| ~ 0.05: a ratio of ~200
    flags, gd_h = run(True, big)
    _, gd_r = run(False, big)
    assert flags is not None and int(flags.sum()) == 0
    assert torch.equal(gd_h, gd_r), 'cancelling planes must take the real dot product'
    small = base * 0.05
    _, gd_h2 = run(True, small)
    _, gd_r2 = run(False, small)
    assert not torch.equal(gd_h2, gd_r2), 'comparable magnitudes stay on the homogeneous route'
    _close_rel(gd_h2, gd_r2, 4e-2 if dtype == torch.bfloat16 else 8e-3, 'homogeneity vs real dot products')


# ---- r06: <x, dx> per plane from the weight gradient's per-image slabs (C ABI afcm_conv2d_wgrad_dots_ld, fused_layer.LayerLink) -----------
@pytest.mark.parametrize('dtype,tol', [(torch.bfloat16, 1.5e-2), (torch.float16, 2e-3)])
@pytest.mark.parametrize('case', [(16, 64, 64, 40, 44, 3), (2, 24, 40, 30, 36, 3), (16, 91, 64, 24, 278, 3), (4, 64, 1, 32, 256, 1), (3, 16, 16, 90, 20, 3), (3, 16, 16, 200, 20, 3)], ids=str)
def test_wgrad_dots_equal_the_plane_dots_of_x_and_dx(case, dtype, tol):
    """dots[n, i] = sum_{o, tap} wq dW_n = <x[n, i], conv^T(wq, dy)[n, i]>: the weight gradient stays what it was (bit for bit: the same K
    order inside an image ... summed over images in a fixed order), the dots match the pixel-side dot products of x with the STORED
    (16-bit) dx up to that rounding; a batch the split count is not a multiple of (n = 3 at 200 rows: 256 shares) has no slab form (None)."""
    from afcm_amd.torch_utils.ops import conv2d as C
    n, cin, cout, h, w, ks = case
    pad = ks - 1
    g = torch.Generator().manual_seed(5)
    x = torch.randn([n, cin, h, w], generator=g).cuda().to(dtype)
    dy = torch.randn([n, cout, h + 2 * pad - ks + 1, w + 2 * pad - ks + 1], generator=g).cuda().to(dtype)
    wt = (torch.randn([cout, cin, ks, ks], generator=g) / np.sqrt(cin * ks * ks)).cuda()
    dw_plain = C._wgrad_raw(dy, x, cout, cin, ks, pad)
    dw, dots = C._wgrad_raw(dy, x, cout, cin, ks, pad, dots_with=wt)
    ref_dw = torch.nn.grad.conv2d_weight(x.double().cpu(), wt.shape, dy.double().cpu(), padding=pad).float()
    _close_rel(dw, ref_dw, 2e-4, 'dw (image-aligned shares)')
    _close_rel(dw_plain, ref_dw, 2e-4, 'dw')
    if h == 200:
        assert dots is None            # 256 shares, three images: no share count per image (the small n = 3 case gets 138 = 3 x 46)
        return
    assert dots is not None and dots.shape == (n, cin) and dots.dtype == torch.float32
    wq = wt.to(dtype).double().cpu()
    dx = torch.nn.grad.conv2d_input(x.shape, wq, dy.double().cpu(), padding=pad)
    want = (x.double().cpu() * dx).sum([2, 3]).float()
    _close_rel(dots, want, 2e-4, 'dots vs float64')
    # ... and against what the product computed before: a pass over x and the stored dx
    wp, rows_pad = C.pack_weights(wt, dtype, 1)
    dx16 = C._conv_raw(dy, wp, rows_pad, None, cin, ks, ks - 1 - pad)
    _close_rel(dots, C.plane_dot(x, dx16), tol, 'dots vs plane_dot(x, stored dx)')


@pytest.mark.parametrize('dtype,tol', [(torch.bfloat16, 4e-2), (torch.float16, 8e-3)])
def test_linked_layers_take_the_style_gradient_from_the_consumers_weight_gradient(dtype, tol):
    """Two fused nodes in a row, the first multiplying its output by the second's styles (next_scale / prescaled), with a LayerLink between them:
    the first node's <g, z> comes from the second node's weight-gradient slabs.  Every gradient against the same chain without the link
    (a pass over g and z), and the link is used (gz delivered and consumed)."""
    from afcm_amd.torch_utils.ops import conv2d as C
    from afcm_amd.torch_utils.ops import fused_layer
    from oracle import generator as ogen
    pl = ogen.plan(256, 4, 1, {})
    L = [l for l in pl['dec'] if l['name'] == 'L12_276_64'][0]
    torch.manual_seed(3)
    n, c0, c1, c2, h = 16, 16, 32, 16, 36
    act = dict(up=2, down=2, padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=256.0)
    fu, fd = L['fu'].cuda(), L['fd'].cuda()
    # (the layer's geometry at a small plane: up 2 / down 2 with its padding keeps the size)
    leaves = dict(x=torch.randn(n, c0, h, h).to(dtype), w1=torch.randn(c1, c0, 3, 3) / np.sqrt(c0 * 9), s1=torch.rand(n, c0) + 0.5, d1=torch.rand(n, c1) + 0.5,
                  b1=torch.randn(c1) * 0.1, w2=torch.randn(c2, c1, 3, 3) / np.sqrt(c1 * 9), s2=torch.rand(n, c1) + 0.5, d2=torch.rand(n, c2) + 0.5, b2=torch.randn(c2) * 0.1)

    def run(linked):
        dev = {k: v.clone().cuda().requires_grad_(True) for k, v in leaves.items()}
        link = fused_layer.LayerLink() if linked else None
        z1 = fused_layer.conv_filtered_lrelu(dev['x'], dev['w1'], dev['s1'], dev['d1'], dev['b1'], fu, fd, conv_pad=2, next_scale=dev['s2'], link_out=link, **act)
        z2 = fused_layer.conv_filtered_lrelu(z1, dev['w2'], dev['s2'], dev['d2'], dev['b2'], fu, fd, conv_pad=2, prescaled=True, link_in=link, **act)
        r = torch.randn(z2.shape, generator=torch.Generator().manual_seed(1)).cuda()
        if linked:
            assert link.want and link.gz is None
        grads = torch.autograd.grad((z2.float() * r).sum(), list(dev.values()))
        return z2, dict(zip(dev.keys(), grads)), link

    z_ref, g_ref, _ = run(False)
    used = []
    orig = C.plane_dot
    C.plane_dot = lambda a, b=None: (used.append(tuple(a.shape)), orig(a, b))[1]
    try:
        z_l, g_l, link = run(True)
    finally:
        C.plane_dot = orig
    assert torch.equal(z_l, z_ref)
    z1_shape = (n, c1, z_ref.shape[2], z_ref.shape[3])
    assert link.gz is None and z1_shape not in used, 'the first node still passed over g and z'
    for k in g_ref:
        _close_rel(g_l[k], g_ref[k], tol if k in ('s2', 'd1') else 1e-5, f'd{k}')


# ---- r06: the direct kernel for convs with at most four input channels (C ABI afcm_conv2d_ld -> conv2d_direct.hip) ---------------------------
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', [(2, 4, 64, 276, 276, 2), (1, 4, 64, 30, 34, 2), (3, 3, 20, 17, 70, 2), (2, 1, 64, 40, 130, 1), (1, 4, 33, 16, 64, 0), (2, 2, 5, 5, 6, 2)], ids=str)
def test_direct_conv_for_few_input_channels(case, dtype):
    """3x3 convs with cin <= 4 (the generator's encoder_0: 4 -> 64 at 276^2) run the direct kernel: every output against float64 aten on the
    16-bit-rounded operands (one rounding of the output), with per-plane factors and a bias, dense and row-pitched outputs, the padding
    columns of a pitched output finite up to the next multiple of 8 (the contract of afcm_conv2d_ld), NaN only where a window holds one."""
    from afcm_amd.torch_utils.ops import _rows
    from afcm_amd.torch_utils.ops import conv2d as C
    n, cin, cout, h, w, pad = case
    g = torch.Generator().manual_seed(h * w + cin)
    x = torch.randn([n, cin, h, w], generator=g).to(dtype)
    wt = torch.randn([cout, cin, 3, 3], generator=g) / np.sqrt(9 * cin)
    osc = torch.rand([n, cout], generator=g) + 0.5
    ob = torch.randn([cout], generator=g) * 0.3
    ref = torch.nn.functional.conv2d(x.double(), wt.to(dtype).double(), padding=pad) * osc.double()[:, :, None, None] + ob.double()[None, :, None, None]
    wp, rows_pad = C.pack_weights(wt.cuda(), dtype, 0)
    eps = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    for pitched in (False, True):
        y = C._conv_raw(x.cuda(), wp, rows_pad, osc.cuda(), cout, 3, pad, obias=ob.cuda(), pitched_out=pitched)
        assert y.dtype == dtype and tuple(y.shape) == tuple(ref.shape)
        err = (y.double().cpu() - ref).abs().max().item()
        assert err <= eps * float(ref.abs().max()) * 1.01, (pitched, err)
        full = _rows.whole_buffer(y)
        if full is not None:
            q = y.shape[3]
            assert bool(torch.isfinite(full[..., q:min(full.shape[3], (q + 7) // 8 * 8)].float()).all())
    xn = x.clone()
    xn[0, 0, h // 2, w // 2] = float('nan')
    yn = C._conv_raw(xn.cuda(), wp, rows_pad, None, cout, 3, pad).float().cpu()
    bad = torch.zeros_like(yn[0, 0], dtype=torch.bool)
    cy, cx = h // 2 + pad, w // 2 + pad                         # output pixels (cy - r, cx - s), r, s in 0..2, see the element
    bad[max(cy - 2, 0):cy + 1, max(cx - 2, 0):cx + 1] = True
    assert bool(yn[0, :, bad].isnan().all()) and bool(torch.isfinite(yn[0][:, ~bad]).all()) and bool(torch.isfinite(yn[1:]).all())
