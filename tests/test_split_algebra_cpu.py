"""CPU checks of the split-operand algebra behind the fp32 convs on the 16-bit matrix pipe (afcm_amd/torch_utils/ops/conv2d.py:
_SPLIT_TERMS, FP32_SPLIT; kernels: csrc/conv2d.hip split16_kernel / conv2d_fwd16_kernel<.., SPLIT>): the term tables and the
power-of-two scaling are emulated with torch's CPU float16 / bfloat16 rounding (round to nearest even, as the kernels' conversions)
and float64 accumulation, so that what remains is exactly the error of the splitting itself."""
import pytest
import torch

from afcm_amd.torch_utils.ops import conv2d as C


def _parts(v, dtype, k):
    out, r = [], v.clone()
    for _ in range(k):
        q = r.to(dtype).to(torch.float32)
        out.append(q)
        r = r - q                      # exact in fp32: q is r rounded to fewer bits
    return out


def _pow2(t):
    e = torch.frexp(t.abs().max().clamp_min(1e-30))[1].item()
    return 2.0 ** (15 - e)


@pytest.mark.parametrize('dtype,terms,bound', [(torch.float16, 3, 2.0 ** -21), (torch.bfloat16, 6, 2.0 ** -22), (torch.bfloat16, 3, 2.0 ** -14)])
@pytest.mark.parametrize('spread', [0.0, 3.0])
def test_term_tables_reach_their_precision(dtype, terms, bound, spread):
    g = torch.Generator().manual_seed(7)
    k = 2304
    x = torch.randn([k, 96], generator=g) * torch.exp(spread * torch.randn([k, 1], generator=g))      # heavy-tailed magnitudes
    w = torch.randn([48, k], generator=g) / k ** 0.5
    ref = w.double() @ x.double()
    gx, gw = (_pow2(x), _pow2(w)) if dtype == torch.float16 else (1.0, 1.0)
    table = C._SPLIT_TERMS[terms]
    nparts = C._nparts(terms)
    assert nparts == (3 if terms == 6 else 2) and len(table) == terms
    xp, wp = _parts(x * gx, dtype, nparts), _parts(w * gw, dtype, nparts)
    assert all(bool(torch.isfinite(p).all()) for p in xp + wp)                       # the scaled float16 parts cannot overflow
    got = sum(wp[b].double() @ xp[a].double() for a, b in table) / (gx * gw)
    # error of the dropped products, relative to the size of the sum of |products| (what an fp32 dot product is measured against)
    size = (w.abs().double() @ x.abs().double())
    assert float(((got - ref).abs() / size).max()) <= bound, float(((got - ref).abs() / size).max())


def test_tables_hold_every_product_up_to_their_order():
    # a term (a, b) multiplies part a of the activations with part b of the weights; part k is ~2^-8k (bf16) / 2^-11k (f16) of the value
    assert sorted(C._SPLIT_TERMS[3]) == [(0, 0), (0, 1), (1, 0)]                                  # order <= 1 of a two-way split
    assert sorted(C._SPLIT_TERMS[6]) == sorted((a, b) for a in range(3) for b in range(3) if a + b <= 2)
    for table in C._SPLIT_TERMS.values():                                                        # smallest products first
        orders = [a + b for a, b in table]
        assert orders == sorted(orders, reverse=True)


def test_default_mode_is_fp32_grade():
    assert C.FP32_SPLIT is None or C.FP32_SPLIT[0] == torch.float16 or min(C.FP32_SPLIT[1:]) >= 6 or C.FP32_SPLIT[1:3] == (6, 6)
