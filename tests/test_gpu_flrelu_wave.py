"""The wave-autonomous matrix-core filtered_lrelu kernels (csrc/filtered_lrelu_wave.hip; 16-bit activations, no bias operand --
the generator's path, where the producing conv adds the bias): one wave per output tile, input fragments loaded straight from
global memory, no LDS staging, composite (slope DV UV) operator, column-blocked row-quad sign codes (sign_layout 2).

Oracle: oracle/aten_ops.filtered_lrelu in fp32 on the same 16-bit-rounded inputs; sign codes additionally against the
definition-level restatement oracle/direct_np (decoded from layout 2) and against the LDS-tile kernels (layout 1)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LAYERS = ['encoder_0', 'encoder_1', 'encoder_4', 'encoder_5', 'encoder_7', 'encoder_9', 'encoder_11', 'encoder_12',
          'L3_52_512', 'L5_84_512', 'L7_148_362', 'L10_276_128', 'L12_276_64', 'L13_256_64']


def _layer(name, res=256):
    from oracle import generator as ogen
    pl = ogen.plan(res, 4, 1, {})
    return [l for l in pl['enc'] + pl['dec'] if l['name'] == name][0]


def _decode_layout2(s, sh_rows):
    """uint8 [N, C, shq, swq] buffer in layout 2 -> codes [N, C, 4 shq, swq].  Byte of quad-row q, column c:
    [c / 16][V / 4][q % 4][c % 16][V % 4] with V = q / 4 (csrc/filtered_lrelu_wave.hip)."""
    n, c, shq, swq = s.shape
    assert shq % 16 == 0 and swq % 16 == 0
    b = s.reshape(n, c, swq // 16, shq // 16, 4, 16, 4)                   # [blk][V4][gq][col in block][V % 4]
    b = np.transpose(b, (0, 1, 3, 6, 4, 2, 5)).reshape(n, c, shq, swq)    # quad-row = (V4 * 4 + V % 4) * 4 + gq; col = blk * 16 + col in block
    codes = np.stack([(b >> (2 * r)) & 3 for r in range(4)], axis=3).reshape(n, c, 4 * shq, swq)
    return codes[:, :, :sh_rows]


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 6e-3), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize('lname', LAYERS)
def test_wave_kernels_forward_backward_vs_oracle(lname, dtype, tol):
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    L = _layer(lname)
    h = L['in_size'] + 2
    torch.manual_seed(5)
    x = torch.randn(2, 3, h, h).to(dtype)
    kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=256.0)
    xr = x.float().requires_grad_(True)
    ref = ops.filtered_lrelu(xr, fu=L['fu'], fd=L['fd'], b=None, **kw)
    r = torch.randn_like(ref).to(dtype)
    gref, = torch.autograd.grad((ref * r.float()).sum(), xr)
    xg = x.cuda().requires_grad_(True)
    got = flr.filtered_lrelu(xg, fu=L['fu'].cuda(), fd=L['fd'].cuda(), b=None, **kw)
    assert got.dtype == dtype and got.shape == ref.shape
    assert got.grad_fn.sign_layout == 2, 'expected the wave-autonomous kernels'
    err = (got.float().cpu() - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f'{lname} {dtype} y: {err:.3e}'
    ggot, = torch.autograd.grad((got.float() * r.cuda().float()).sum(), xg)
    rel = ((ggot.float().cpu() - gref).norm() / gref.norm()).item()       # 16-bit rounding flips leaky-ReLU branches near 0
    assert rel <= 2 * tol, f'{lname} {dtype} dx: relative L2 {rel:.3e}'
    # inference mode (no sign tensor) runs the same arithmetic
    with torch.no_grad():
        y2 = flr.filtered_lrelu(x.cuda(), fu=L['fu'].cuda(), fd=L['fd'].cuda(), b=None, **kw)
    assert torch.equal(y2, got.detach())


@pytest.mark.parametrize('lname', ['encoder_1', 'encoder_4', 'L3_52_512', 'encoder_12'])
def test_wave_kernels_sign_codes(lname):
    """Codes of the region every tile owns, decoded from layout 2, vs the definition-level restatement (wherever the
    pre-activation is not within 16-bit rounding distance of 0)."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import direct_np as dnp
    L = _layer(lname)
    h = L['in_size'] + 2
    torch.manual_seed(3)
    x = torch.randn(2, 2, h, h).to(torch.float16)
    kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=256.0)
    xg = x.cuda().requires_grad_(True)
    y = flr.filtered_lrelu(xg, fu=L['fu'].cuda(), fd=L['fd'].cuda(), b=None, **kw)
    assert y.grad_fn.sign_layout == 2
    s = y.grad_fn.saved_tensors[2].cpu().numpy()
    u = dnp.upfirdn2d(x.float().numpy().astype(np.float64), L['fu'].numpy(), up=L['up'], padding=L['padding'], gain=float(L['up'] ** 2),
                      flip_filter=False)
    _, want = dnp.lrelu_codes(u, kw['gain'], kw['slope'], kw['clamp'])
    sh = L['out_size'] * L['down'] - (L['down'] - 1) + len(L['fd']) - 1
    got = _decode_layout2(s, sh)
    w = min(want.shape[3], got.shape[3])
    uu = u[:, :, :sh, :w]
    safe = (np.abs(uu) > 4e-3 * max(1.0, np.abs(u).max())) | (uu == 0)      # exact zeros (padding) must read code 0
    assert safe.mean() > 0.9
    assert np.array_equal(got[:, :, :, :w][safe], want[:, :, :sh, :w][safe])


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 6e-3), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize('lname', ['encoder_1', 'encoder_4', 'L10_276_128', 'encoder_11', 'L3_52_512'])
def test_wave_kernels_clamp(lname, dtype, tol):
    """Inputs scaled so that the clamp is reached in some column blocks and not in others: the per-block exact path of the
    forward kernel (max |X1| bound) and the clamp-code path of the transposed kernel (gradient 0 where clamped)."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    L = _layer(lname)
    h = L['in_size'] + 2
    torch.manual_seed(11)
    x = torch.randn(2, 2, h, h)
    x[0, 0] *= 40.0
    x[1, 1, : h // 2] *= 12.0
    x[1, 0, :, : h // 3] *= 300.0        # far above the clamp: no cancellation between the linear and the relu operand allowed
    x = x.to(dtype)
    kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=8.0)
    xr = x.float().requires_grad_(True)
    ref = ops.filtered_lrelu(xr, fu=L['fu'], fd=L['fd'], b=None, **kw)
    r = torch.randn_like(ref).to(dtype)
    gref, = torch.autograd.grad((ref * r.float()).sum(), xr)
    xg = x.cuda().requires_grad_(True)
    got = flr.filtered_lrelu(xg, fu=L['fu'].cuda(), fd=L['fd'].cuda(), b=None, **kw)
    assert got.grad_fn.sign_layout == 2
    err = (got.float().cpu() - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f'{lname} {dtype} clamp y: {err:.3e}'
    ggot, = torch.autograd.grad((got.float() * r.cuda().float()).sum(), xg)
    rel = ((ggot.float().cpu() - gref).norm() / gref.norm()).item()
    assert rel <= 3 * tol, f'{lname} {dtype} clamp dx: relative L2 {rel:.3e}'


# r06: column groups whose X3 pair no output of the plane reads run the down-x pass only (tail_group, csrc/filtered_lrelu_wave.hip).  Their number
# depends on the plane's width modulo 16 (the generator's planes are all 16 m + 4 wide: one such group per strip): every residue class, incl. the
# widths with NO such group (residues 12 .. 16) and the down-4 widths with TWO (residues <= 3), forward (codes written) and transposed (codes read).
@pytest.mark.parametrize('lname,h', [('encoder_1', 20), ('encoder_1', 24), ('encoder_1', 28), ('encoder_1', 30), ('encoder_1', 32), ('encoder_1', 34),
                                     ('encoder_1', 46), ('encoder_1', 66),
                                     ('encoder_4', 18), ('encoder_4', 22), ('encoder_4', 38), ('encoder_4', 50), ('encoder_4', 54),
                                     ('L3_52_512', 18), ('L3_52_512', 20), ('L3_52_512', 22), ('L3_52_512', 26), ('L3_52_512', 30)])
def test_wave_kernels_plane_width_sweep(lname, h):
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    from oracle import aten_ops as ops
    from oracle import direct_np as dnp
    L = _layer(lname)
    dtype, tol = torch.float16, 6e-3
    torch.manual_seed(h)
    x = torch.randn(2, 3, h, h).to(dtype)
    kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=float(np.sqrt(2)), slope=0.2, clamp=256.0)
    xr = x.float().requires_grad_(True)
    ref = ops.filtered_lrelu(xr, fu=L['fu'], fd=L['fd'], b=None, **kw)
    assert ref.shape[3] % 2 == 0, 'even plane widths only (matrix-core kernels)'
    r = torch.randn_like(ref).to(dtype)
    gref, = torch.autograd.grad((ref * r.float()).sum(), xr)
    xg = x.cuda().requires_grad_(True)
    got = flr.filtered_lrelu(xg, fu=L['fu'].cuda(), fd=L['fd'].cuda(), b=None, **kw)
    assert got.shape == ref.shape and got.grad_fn.sign_layout == 2, 'expected the wave-autonomous kernels'
    err = (got.float().cpu() - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f'{lname} {h} (width {ref.shape[3]}) y: {err:.3e}'
    s = got.grad_fn.saved_tensors[2].cpu().numpy()                      # (before the backward pass frees it)
    ggot, = torch.autograd.grad((got.float() * r.cuda().float()).sum(), xg)
    rel = ((ggot.float().cpu() - gref).norm() / gref.norm()).item()
    assert rel <= 2 * tol, f'{lname} {h} (width {ref.shape[3]}) dx: relative L2 {rel:.3e}'
    # every code of the sign tensor, also in the column blocks beyond the last column an output reads
    u = dnp.upfirdn2d(x.float().numpy().astype(np.float64), L['fu'].numpy(), up=L['up'], padding=L['padding'], gain=float(L['up'] ** 2), flip_filter=False)
    _, want = dnp.lrelu_codes(u, kw['gain'], kw['slope'], kw['clamp'])
    sh = min(want.shape[2], 4 * s.shape[2])
    codes = _decode_layout2(s, sh)
    w = min(want.shape[3], codes.shape[3])
    uu = u[:, :, :sh, :w]
    safe = (np.abs(uu) > 4e-3 * max(1.0, np.abs(u).max())) | (uu == 0)
    assert safe.mean() > 0.9
    assert np.array_equal(codes[:, :, :, :w][safe], want[:, :, :sh, :w][safe])


def test_wave_and_lds_tile_kernels_agree(monkeypatch):
    """Same call through both matrix-core families (the LDS-tile family is reached through its bias operand: b = 0): outputs and input gradients agree to 16-bit rounding; epilogue operands (skip, per-plane
    factors, per-tile output sums) behave identically."""
    from afcm_amd.torch_utils.ops import filtered_lrelu as flr
    L = _layer('L9_148_181')
    h = L['in_size'] + 2
    torch.manual_seed(2)
    n, c = 2, 5
    x = torch.randn(n, c, h, h, device='cuda', dtype=torch.bfloat16)
    fu, fd = L['fu'].cuda(), L['fd'].cuda()
    cfg = (L['up'], L['down'], *L['padding'], float(np.sqrt(2)), 0.2, 256.0, False, 0, 0, 0)
    skip = torch.randn(n, c, L['out_size'], L['out_size'], device='cuda', dtype=torch.bfloat16)
    osc = torch.rand(n * c, device='cuda') + 0.5
    y_w, s_w, lay_w, _ = flr._run(x, fu, fd, None, None, cfg, True, oscale=osc, skip=skip)
    y_l, s_l, lay_l, _ = flr._run(x, fu, fd, torch.zeros(c, device='cuda', dtype=torch.bfloat16), None, cfg, True, oscale=osc, skip=skip)
    assert (lay_w, lay_l) == (2, 1)
    assert (y_w.float() - y_l.float()).abs().max().item() <= 4e-2 * y_l.float().abs().max().item()
    dy = torch.randn_like(y_w)
    bcfg = flr._backward_cfg(cfg, fu, fd, x.shape, y_w.shape, lay_w)
    g_w, _, _, ps_w = flr._run(dy, fd, fu, None, s_w, bcfg, False, want_plane_sum=True)
    bcfg_l = flr._backward_cfg(cfg, fu, fd, x.shape, y_l.shape, lay_l)
    g_l, _, _, ps_l = flr._run(dy, fd, fu, None, s_l, bcfg_l, False, want_plane_sum=True)
    assert ((g_w.float() - g_l.float()).norm() / g_l.float().norm()).item() <= 2e-2
    assert ps_w is not None and ps_l is not None           # (slots per plane differ: strips vs tiles)
    want = g_w.float().sum([2, 3])
    assert (ps_w.sum(2) - want).abs().max().item() <= 2e-2 * max(1.0, want.abs().max().item())
