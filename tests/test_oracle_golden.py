"""Pin the CPU oracle to golden vectors captured from the real reference (tools/gen_golden.py).

Tolerance: <= 1e-5 max-abs relative to the output scale (fp32 summation-order noise only).
"""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import aten_ops as ops
from oracle import direct_np as dnp
from oracle import generator as ogen

TOL = 1e-5


def _t(a, grad=False):
    if a is None:
        return None
    t = torch.from_numpy(np.array(a))
    return t.requires_grad_(True) if grad else t


def _close(a, b, tol=TOL, what=''):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f'{what}: max-abs err {err:.3e} (scale {scale:.3g})'


@pytest.mark.parametrize('name', golden_names('F'))
def test_filtered_lrelu(name):
    g = load_golden(name)
    up, down, *pad = [int(v) for v in g['meta']]
    gain, slope, clamp, flip = g['fmeta']
    clamp = None if clamp < 0 else float(clamp)
    x = _t(g['x'], True)
    b = _t(g.get('b'), True)
    y = ops.filtered_lrelu(x, fu=_t(g.get('fu')), fd=_t(g.get('fd')), b=b, up=up, down=down, padding=pad, gain=float(gain),
                           slope=float(slope), clamp=clamp, flip_filter=bool(flip))
    _close(y, g['y'], what=name + ' y')
    grads = torch.autograd.grad((y * _t(g['r'])).sum(), [x] + ([b] if b is not None else []))
    _close(grads[0], g['dx'], what=name + ' dx')
    if b is not None:
        _close(grads[1], g['db'], what=name + ' db', tol=1e-4)
    # independent definition-level restatement (float64)
    y2 = dnp.filtered_lrelu(g['x'], g.get('fu'), g.get('fd'), g.get('b'), up, down, pad, float(gain), float(slope), clamp, bool(flip))
    _close(y2, g['y'], what=name + ' y(direct)', tol=2e-5)


@pytest.mark.parametrize('name', golden_names('U'))
def test_upfirdn2d(name):
    g = load_golden(name)
    up, down, px0, px1, py0, py1, flip = [int(v) for v in g['meta']]
    gain = float(g['fmeta'][0])
    fn = str(g['fn'])
    x = _t(g['x'], True)
    f = _t(g['f'])
    if fn == 'upfirdn2d':
        y = ops.upfirdn2d(x, f, up=up, down=down, padding=[px0, px1, py0, py1], flip_filter=bool(flip), gain=gain)
        y2 = dnp.upfirdn2d(g['x'], g['f'], up, down, [px0, px1, py0, py1], bool(flip), gain)
        _close(y2, g['y'], what=name + ' y(direct)', tol=2e-5)
    elif fn == 'filter2d':
        y = ops.filter2d(x, f, padding=[px0, px1, py0, py1], flip_filter=bool(flip), gain=gain)
    elif fn == 'upsample2d':
        y = ops.upsample2d(x, f, up=up, padding=[px0, px1, py0, py1], flip_filter=bool(flip), gain=gain)
    else:
        y = ops.downsample2d(x, f, down=down, padding=[px0, px1, py0, py1], flip_filter=bool(flip), gain=gain)
    _close(y, g['y'], what=name + ' y')
    dx, = torch.autograd.grad((y * _t(g['r'])).sum(), [x])
    _close(dx, g['dx'], what=name + ' dx')


@pytest.mark.parametrize('name', golden_names('B'))
def test_bias_act(name):
    g = load_golden(name)
    alpha, gain, clamp = [None if np.isnan(v) else float(v) for v in g['fmeta']]
    x = _t(g['x'], True)
    b = _t(g.get('b'), True)
    y = ops.bias_act(x, b, dim=int(g['dim']), act=str(g['act']), alpha=alpha, gain=gain, clamp=clamp)
    _close(y, g['y'], what=name)
    grads = torch.autograd.grad((y * _t(g['r'])).sum(), [x, b])
    _close(grads[0], g['dx'], what=name + ' dx')
    _close(grads[1], g['db'], what=name + ' db')


@pytest.mark.parametrize('name', golden_names('M'))
def test_modulated_conv2d(name):
    g = load_golden(name)
    demod, padding = [int(v) for v in g['meta']]
    ig = None if np.isnan(g['fmeta'][0]) else torch.tensor(float(g['fmeta'][0]))
    x, w, s = _t(g['x'], True), _t(g['w'], True), _t(g['s'], True)
    y = ops.modulated_conv2d(x, w, s, demodulate=bool(demod), padding=padding, input_gain=ig)
    _close(y, g['y'], what=name)
    dx, dw, ds = torch.autograd.grad((y * _t(g['r'])).sum(), [x, w, s])
    _close(dx, g['dx'], what=name + ' dx')
    _close(dw, g['dw'], what=name + ' dw', tol=1e-4)
    _close(ds, g['ds'], what=name + ' ds', tol=1e-4)


TINY = dict(channel_base=256, channel_max=8)


@pytest.mark.parametrize('name,res', [('G1_tiny128', 128), ('G2_tiny256', 256), ('G3_tiny512', 512)])
def test_generator(name, res):
    g = load_golden(name)
    sd = {k[3:]: _t(v) for k, v in g.items() if k.startswith('sd/')}
    pl = ogen.plan(res, 4, 1, dict(TINY, channel_base=1024) if res == 512 else TINY)
    # the plan must reproduce the reference's layer names and the filters stored in its state dict
    names = [L['name'] for L in pl['enc'] + pl['dec']]
    assert names == [str(n) for n in g['layer_names']]
    for L in pl['enc'] + pl['dec']:
        for key, f in (('up_filter', L['fu']), ('down_filter', L['fd'])):
            k = f'synthesis.{L["name"]}.{key}'
            assert (k in sd) == (f is not None)
            if f is not None:
                _close(f, sd[k].numpy(), tol=1e-7, what=k)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith(('_filter', 'magnitude_ema', 'w_avg'))}
    full = dict(sd)
    full.update(params)
    taps = {}
    y = ogen.generator(full, pl, _t(g['z']), _t(g['c']), _t(g['x']), mapping_layers=2, taps=taps)
    _close(y, g['y'], what=name + ' y', tol=2e-5)
    for lname, t in taps.items():
        st = g['stat/' + lname]
        got = np.array([t.mean().item(), t.std().item(), t.abs().max().item()])
        assert np.allclose(got, st, rtol=1e-3, atol=1e-5), (lname, got, st)
    want = {k[5:]: v for k, v in g.items() if k.startswith('grad/')}
    names = list(want.keys())
    grads = torch.autograd.grad((y * _t(g['r'])).sum(), [params[k] for k in names])
    for k, gr in zip(names, grads):
        _close(gr, want[k], tol=1e-4, what=name + ' grad ' + k)


def test_layer_table_full_width():
    """Geometry of the shipped 256^2 configuration (SURVEY.md section 8 layer table)."""
    g = load_golden('T256_layer_table')
    pl = ogen.plan(256, 4, 1, {})
    layers = pl['enc'] + pl['dec']
    assert [L['name'] for L in layers] == [str(n) for n in g['names']]
    for L, row in zip(layers, g['table']):
        cin, cout, insz, outsz, up, down, ut, dt, p0, p1, p2, p3, k = [int(v) for v in row]
        assert (L['cin'], L['cout'], L['in_size'], L['out_size'], L['up'], L['down'], L['k']) == (cin, cout, insz, outsz, up, down, k)
        assert L['padding'] == [p0, p1, p2, p3]
        assert (1 if L['fu'] is None else len(L['fu'])) == ut
        assert (1 if L['fd'] is None else len(L['fd'])) == dt
        for key, f in (('fu/', L['fu']), ('fd/', L['fd'])):
            if f is not None:
                _close(f, g[key + L['name']], tol=1e-7, what=key + L['name'])
    sd = ogen.random_state_dict(pl, 512, 1, 512, 8)
    assert sum(v.numel() for k, v in sd.items() if not k.endswith(('magnitude_ema', 'w_avg'))) == int(g['nparams'])


def test_sign_code_packing():
    codes = np.random.RandomState(0).randint(0, 3, size=(1, 2, 5, 37)).astype(np.uint8)
    packed = dnp.pack_codes_rowmajor(codes)
    assert packed.shape == (1, 2, 5, 12)
    for x in range(37):
        assert np.array_equal((packed[..., x >> 2] >> (2 * (x & 3))) & 3, codes[..., x])


@pytest.mark.parametrize('name', ['D1_tiny64', 'D2_tiny128_clamp', 'D3_tiny64_cond'])
def test_discriminator_oracle_matches_reference(name):
    """Row f1: the functional discriminator restatement (oracle/discriminator.py) vs vectors captured from the reference's
    CoModDiscriminator -- logits, both D loss gradients (incl. the R1 double backward) and the image gradient of the G term.
    D3: the conditional form (c_dim = 1, configs/adni/stylegan3/cmsr.yml:13): label mapping network + projection."""
    import torch
    from oracle import discriminator as od
    g = load_golden(name)
    res, n, _, _, group, clamp = [int(v) for v in g['meta']]
    kw = dict(mbstd_group_size=group, conv_clamp=None if clamp < 0 else float(clamp))
    if 'c' in g:
        kw['c'] = torch.from_numpy(g['c'])
    names = [str(k) for k in g['names']]
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith('sd/')}
    for k in names:
        sd[k] = sd[k].clone().requires_grad_(True)
    fake, real = torch.from_numpy(g['fake']), torch.from_numpy(g['real'])
    lf, lr, l1, gen_logits, real_logits, r1 = od.d_losses(sd, fake, real, res, **kw)
    assert np.allclose(gen_logits.detach().numpy(), g['gen_logits'], atol=1e-5)
    assert np.allclose(real_logits.detach().numpy(), g['real_logits'], atol=1e-5)
    assert np.allclose(r1.detach().numpy(), g['r1_grads'], atol=1e-6 + 1e-4 * np.abs(g['r1_grads']).max())
    assert abs(lf.item() - float(g['loss_fake'])) < 1e-5 and abs(l1.item() - float(g['loss_r1'])) < 1e-6 + 1e-4 * float(g['loss_r1'])
    gf = torch.autograd.grad(lf, [sd[k] for k in names], retain_graph=True)
    gr = torch.autograd.grad(lr + l1 * 10.0, [sd[k] for k in names], retain_graph=True)
    for k, a, b in zip(names, gf, gr):
        for got, want, what in ((a, g['gfake/' + k], 'fake'), (b, g['greal/' + k], 'real+r1')):
            tol = 1e-5 * max(1.0, float(np.abs(want).max()))
            assert np.abs(got.numpy() - want).max() <= tol, (name, k, what)
    # the R1 double backward on its own, relative to ITS OWN scale (it is a few percent of greal)
    g1 = torch.autograd.grad(l1, [sd[k] for k in names], allow_unused=True)
    r1_scale = max(float(np.abs(g['gr1/' + k]).max()) for k in names)
    for k, a in zip(names, g1):
        want = g['gr1/' + k]
        got = np.zeros_like(want) if a is None else a.numpy()
        # fp32 on both sides: bias gradients of the penalty are sums of cancelling terms (|sum| ~ 1e-6 of terms ~ 1e-3), whose
        # last bits depend on the summation order -> a floor of 1e-5 of the largest R1 gradient next to the 1e-3 relative bound
        assert np.abs(got - want).max() <= 1e-3 * float(np.abs(want).max()) + 1e-5 * r1_scale, (name, k, 'r1')
    img = fake.clone().requires_grad_(True)
    lg = torch.nn.functional.softplus(-od.discriminator(sd, img, res, **kw)).mean()
    gi, = torch.autograd.grad(lg, img)
    assert np.abs(gi.numpy() - g['g_img']).max() <= 1e-5 * max(1e-3, float(np.abs(g['g_img']).max()))


def test_resampling_conv2dlayer_oracle_matches_reference():
    """The reference's Conv2dLayer with up / down (CoModGAN/layers.py:115-162) for every branch of conv2d_resample, vs the oracle --
    which computes the up cases from the definition (zero-insert upsample -> filter -> convolution), not from the reference's
    transposed-convolution decomposition."""
    import torch
    from oracle import discriminator as od
    g = load_golden('C1_conv2dlayer_resample')
    filt = od.setup_filter([1, 3, 3, 1])
    for n, (k, up, down) in enumerate(g['cases']):
        k, up, down = int(k), int(up), int(down)
        sd = {'weight': torch.from_numpy(g[f'{n}/w']).requires_grad_(True), 'bias': torch.from_numpy(g[f'{n}/b']).requires_grad_(True)}
        x = torch.from_numpy(g[f'{n}/x']).requires_grad_(True)
        y = od.conv2d_layer(sd, '', x, k, act='lrelu', down=down, up=up, gain=0.7, conv_clamp=(2.0 if n == 0 else None), filt=filt)
        assert y.shape == g[f'{n}/y'].shape, (k, up, down)
        assert np.abs(y.detach().numpy() - g[f'{n}/y']).max() <= 1e-5, (k, up, down)
        gx, gw, gb = torch.autograd.grad((y * torch.from_numpy(g[f'{n}/r'])).sum(), [x, sd['weight'], sd['bias']])
        for got, key in ((gx, 'gx'), (gw, 'gw'), (gb, 'gb')):
            want = g[f'{n}/{key}']
            assert np.abs(got.numpy() - want).max() <= 1e-5 * max(1.0, float(np.abs(want).max())), (k, up, down, key)


def test_filtered_lrelu_branch_override_is_self_consistent():
    """oracle.aten_ops.filtered_lrelu(codes=...) with the oracle's OWN branch decisions reproduces value and gradient of the plain
    call (the instrumentation the GPU gradient-attribution test relies on)."""
    import torch
    from oracle import aten_ops as ops
    g = load_golden('F6_clamp')
    up, down, *pad = [int(v) for v in g['meta']]
    gain, slope, clamp, flip = [float(v) for v in g['fmeta']]
    kw = dict(fu=torch.from_numpy(g['fu']), fd=torch.from_numpy(g['fd']), b=torch.from_numpy(g['b']), up=up, down=down, padding=pad,
              gain=gain, slope=slope, clamp=clamp, flip_filter=bool(flip))
    x = torch.from_numpy(g['x']).requires_grad_(True)
    rec = []
    y = ops.filtered_lrelu(x, record=rec, **kw)
    u = rec[0]
    v = u * torch.where(u < 0, slope, 1.0) * gain
    codes = (u < 0).to(torch.uint8) | ((v.abs() > clamp).to(torch.uint8) << 1)
    assert int((codes & 2).ne(0).sum()) > 0
    y2 = ops.filtered_lrelu(x, codes=codes, **kw)
    r = torch.from_numpy(g['r'])
    g1, = torch.autograd.grad((y * r).sum(), x)
    g2, = torch.autograd.grad((y2 * r).sum(), x)
    assert torch.allclose(y, y2, atol=1e-6) and torch.allclose(g1, g2, atol=1e-6)
    assert np.abs(g2.numpy() - g['dx']).max() <= 1e-5 * max(1.0, np.abs(g['dx']).max())
