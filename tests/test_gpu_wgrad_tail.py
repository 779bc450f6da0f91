"""Weight gradient on rows that end in a short chunk (afcm_amd/csrc/conv2d.hip, conv2d_wgrad16g_kernel).

Every generator plane is 64 k + 22 pixels wide: the last 64-pixel chunk of a row pair is a step of its own in which half the waves idle.
The cases walk the step order's corners -- odd and even counts of row pairs, an odd row count (the last pair holds one row), tails of
6 / 22 / 24 / 26 pixels, a row shorter than one chunk, several images, partial channel tiles, and, with 64 x 64 channels, one split per
few steps, so that every workgroup decodes its first step from an arbitrary position of the order.  They were written for the shared tail
steps of profiles/r05_wgrad_tailmerge_experiment.txt (two row pairs in one tail step; measured, not kept) and hold for any step order.
Reference: float64 autograd of F.conv2d on the CPU (the weight gradient the reference's conv2d_gradfix computes,
torch_utils/ops/conv2d_gradfix.py:131-160).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(dy, x, pad):
    w = torch.zeros(dy.shape[1], x.shape[1], 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().cpu(), w, padding=pad)
    y.backward(dy.double().cpu())
    return w.grad


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,cin,cout,h,w', [
    (2, 64, 64, 7, 84),      # Q = 86, P = 9: 5 row pairs (odd), the last holds one row; 256 splits of a few steps
    (1, 64, 64, 10, 84),     # 6 row pairs
    (3, 40, 72, 4, 148),     # Q = 150, 3 row pairs, partial channel tiles
    (1, 64, 128, 2, 276),    # Q = 278, P = 4: exactly one pair of row pairs
    (2, 64, 64, 1, 68),      # Q = 70 (tail 6), P = 3: 2 row pairs, the second with one row
    (2, 32, 64, 5, 86),      # Q = 88: tail 24, the widest that merges
    (2, 32, 64, 5, 88),      # Q = 90: tail 26, not merged
    (1, 64, 64, 1, 20),      # Q = 22: no full chunk, not merged
])


def test_wgrad_tail_steps_match_float64(n, cin, cout, h, w, dtype):
    from afcm_amd.torch_utils.ops import conv2d as conv
    torch.manual_seed(n * 1000 + h * 10 + w)
    x = torch.randn(n, cin, h, w, device='cuda').to(dtype)
    dy = torch.randn(n, cout, h + 2, w + 2, device='cuda').to(dtype)
    got = conv._wgrad_raw(dy, x, cout, cin, 3, 2).double().cpu()
    want = _ref(dy, x, 2)
    assert got.shape == want.shape
    # fp32 accumulation of exact 16-bit products over k = n (h + 2)(w + 2) terms of unit variance: rounding ~ 1e-7 k^0.5 log k; one pixel
    # missed, doubled or paired with the wrong neighbour is an error of order 1
    k = n * (h + 2) * (w + 2)
    err = (got - want).abs().max().item()
    assert err <= 1e-4 * k ** 0.5, (err, 1e-4 * k ** 0.5)


def test_wgrad_tail_steps_are_position_exact():
    """One non-zero dy pixel and one non-zero x pixel at a time: the gradient must be their product at exactly one tap -- the rows of
    neighbouring row pairs and the columns next to the chunk borders and the row's end."""
    from afcm_amd.torch_utils.ops import conv2d as conv
    n, cin, cout, h, w = 1, 16, 16, 6, 84          # P = 8: 4 row pairs; Q = 86 = 64 + 22
    dtype = torch.bfloat16
    for (p, q) in [(0, 64), (1, 85), (2, 64), (3, 85), (3, 70), (4, 63), (5, 64), (6, 85), (7, 64), (7, 0)]:
        for (dr, ds) in [(0, 0), (2, 2), (1, 0), (0, 2)]:
            iy, ix = p + dr - 2, q + ds - 2
            if not (0 <= iy < h and 0 <= ix < w):
                continue
            x = torch.zeros(n, cin, h, w, device='cuda', dtype=dtype)
            dy = torch.zeros(n, cout, h + 2, w + 2, device='cuda', dtype=dtype)
            dy[0, 3, p, q] = 2.0
            x[0, 5, iy, ix] = 3.0
            got = conv._wgrad_raw(dy, x, cout, cin, 3, 2)
            want = torch.zeros_like(got)
            want[3, 5, dr, ds] = 6.0
            assert torch.equal(got, want), (p, q, dr, ds, got.nonzero().tolist())
