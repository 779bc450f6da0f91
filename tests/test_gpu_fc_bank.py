"""The one-launch dense layers (C ABI afcm_fc_act_fwd / _bwd, afcm_mapping_input_fwd / _bwd) against float64 aten and against the op-by-op
composition they replace (NET:69-104, 109-164)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f'{what}: {err:.3e} of scale {scale:.3e} (tolerance {tol:.1e})'


@pytest.mark.parametrize('n,cin,cout,act,bias', [(16, 1024, 512, 'lrelu', True), (16, 512, 512, 'lrelu', True), (2, 4608, 1024, 'lrelu', True),
                                                 (32, 512, 512, 'lrelu', True), (5, 64, 48, 'linear', True), (64, 32, 16, 'linear', False),
                                                 (33, 48, 80, 'lrelu', True), (1, 16, 16, 'lrelu', False), (17, 1536, 512, 'linear', True)], ids=str)
def test_fc_act_forward_backward_vs_float64(n, cin, cout, act, bias):
    from afcm_amd.torch_utils.ops import fc_bank
    g = torch.Generator().manual_seed(cin + cout + n)
    x = torch.randn([n, cin], generator=g)
    w = torch.randn([cout, cin], generator=g)
    b = torch.randn([cout], generator=g) if bias else None
    r = torch.randn([n, cout], generator=g)
    alpha, beta = 0.37 / np.sqrt(cin), 0.71
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if bias else None
    yd = xd @ (wd * alpha).t() + (bd * beta if bias else 0.0)
    if act == 'lrelu':
        yd = torch.nn.functional.leaky_relu(yd, 0.2) * np.sqrt(2)
    ref = torch.autograd.grad((yd * r.double()).sum(), [xd, wd] + ([bd] if bias else []))
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if bias else None
    assert fc_bank.supported(xg, wg, act)
    y = fc_bank.fc_act(xg, wg, bg, alpha, beta, act)
    got = torch.autograd.grad((y * r.cuda()).sum(), [xg, wg] + ([bg] if bias else []))
    _close(y, yd, 2e-6, 'y')
    for nm, a, e in zip(('dx', 'dw', 'db'), got, ref):
        _close(a, e, 3e-6, nm)
    # only some gradients wanted: the others are not computed, the wanted ones do not change
    xg2 = x.cuda().requires_grad_(True)
    y2 = fc_bank.fc_act(xg2, w.cuda(), None if b is None else b.cuda(), alpha, beta, act)
    dx2, = torch.autograd.grad((y2 * r.cuda()).sum(), [xg2])
    assert torch.equal(dx2, got[0]) and torch.equal(y2, y)


def test_fc_act_keeps_nan_and_rejects_unsupported_shapes():
    from afcm_amd.torch_utils.ops import fc_bank
    x = torch.randn(4, 32).cuda()
    w = torch.randn(16, 32).cuda()
    x[1, 3] = float('nan')
    y = fc_bank.fc_act(x, w, None, 1.0, 1.0, 'lrelu')
    assert bool(y[1].isnan().all()) and bool(torch.isfinite(y[[0, 2, 3]]).all())
    assert not fc_bank.supported(torch.randn(4, 30).cuda(), torch.randn(16, 30).cuda(), 'lrelu')       # cin % 16
    assert not fc_bank.supported(torch.randn(65, 32).cuda(), w, 'lrelu')                               # more than 64 rows
    assert not fc_bank.supported(x, w, 'sigmoid') and not fc_bank.supported(x.half(), w, 'lrelu')
    with pytest.raises(RuntimeError):
        fc_bank.fc_act(torch.randn(4, 32), torch.randn(16, 32), None, 1.0, 1.0, 'lrelu')             # no CPU path


@pytest.mark.parametrize('n,cdim', [(16, 1), (3, 4), (5, 0)])
def test_mapping_network_matches_the_op_by_op_composition(n, cdim):
    """MappingNetwork (NET:109-164) with the one-launch kernels against the same module run op by op (GEMM library + bias_act), outputs
    and every parameter gradient; and against a float64 restatement of NET:143-157."""
    from afcm_amd.networks_stylegan3 import MappingNetwork
    from afcm_amd.torch_utils.ops import fc_bank
    torch.manual_seed(n + cdim)
    m = MappingNetwork(z_dim=512, c_dim=cdim, w_dim=512, num_ws=16, num_layers=8).cuda()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.05)        # biases are zero at init
    z, c = torch.randn(n, 512).cuda(), (torch.rand(n, cdim).cuda() if cdim else None)
    r = torch.randn(n, 16, 512).cuda()
    outs = {}
    for mode in (True, False):
        fc_bank.ENABLED = mode
        try:
            ws = m(z, c)
            grads = torch.autograd.grad((ws * r).sum(), list(m.parameters()))
        finally:
            fc_bank.ENABLED = True
        outs[mode] = (ws, grads)
    _close(outs[True][0], outs[False][0], 2e-5, 'ws')
    for (nm, _), a, b in zip(m.named_parameters(), outs[True][1], outs[False][1]):
        _close(a, b, 5e-5, nm)
    # float64 restatement
    sd = {k: v.detach().double().cpu().requires_grad_(True) for k, v in m.named_parameters()}
    x = z.double().cpu()
    x = x * (x.square().mean(1, keepdim=True) + 1e-8).rsqrt()
    if cdim:
        y = c.double().cpu() @ (sd['embed.weight'] * (1 / np.sqrt(cdim))).t() + sd['embed.bias']
        y = y * (y.square().mean(1, keepdim=True) + 1e-8).rsqrt()
        x = torch.cat([x, y], 1)
    for i in range(8):
        w, b = sd[f'fc{i}.weight'], sd[f'fc{i}.bias']
        x = torch.nn.functional.leaky_relu(x @ (w * (0.01 / np.sqrt(w.shape[1]))).t() + b * 0.01, 0.2) * np.sqrt(2)
    wd = x.unsqueeze(1).repeat(1, 16, 1)
    gd = torch.autograd.grad((wd * r.double().cpu()).sum(), list(sd.values()))
    _close(outs[True][0], wd, 1e-5, 'ws vs float64')
    for nm, a, b in zip(sd.keys(), outs[True][1], gd):
        _close(a, b, 3e-5, nm + ' vs float64')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 5, 20, 276), (1, 3, 84, 84), (2, 2, 6, 38)], ids=str)
def test_skip_fork_backward_is_the_scaled_sum_of_both_arms(shape, dtype):
    """_SkipFork (fused_layer.py): x -> (x, x) whose backward is ga + scale * gb in one pass (C ABI afcm_axpy_planes) over row-pitched or dense
    gradients; without a pending scale a plain sum; one arm unused: the other arm's gradient."""
    from afcm_amd.torch_utils.ops import _rows, fused_layer
    n, c, h, w = shape
    g = torch.Generator().manual_seed(w)
    x = _rows.empty(shape, dtype, 'cuda')
    x.copy_(torch.randn(shape, generator=g))
    x.requires_grad_(True)
    ga, gb = _rows.empty(shape, dtype, 'cuda'), _rows.empty(shape, dtype, 'cuda')
    ga.copy_(torch.randn(shape, generator=g)); gb.copy_(torch.randn(shape, generator=g))
    sc = (torch.rand(n, c, generator=g) + 0.5).cuda()
    for scale in (sc, None):
        a, b = fused_layer.skip_fork(x)
        assert a.data_ptr() == x.data_ptr() and b.stride() == x.stride() and hasattr(b, '_afcm_fork')
        b._afcm_fork.scale = scale
        gx, = torch.autograd.grad([a, b], [x], [ga, gb])
        want = ga.float() + gb.float() * (1.0 if scale is None else scale[:, :, None, None])
        # (one rounding on the fused path; two -- the scaled arm, then the sum -- where the plane size keeps the op-by-op route: 6 x 38)
        assert gx.dtype == dtype and float((gx.float() - want).abs().max()) <= float(want.abs().max()) * 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
        assert b._afcm_fork.scale is None                                   # consumed
    a, b = fused_layer.skip_fork(x)
    gx, = torch.autograd.grad([a], [x], [ga])
    assert torch.equal(gx, ga)
    with torch.no_grad():
        a, b = fused_layer.skip_fork(x)
        assert a is x and b is x
