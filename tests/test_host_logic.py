"""CPU tests of the product's host-side logic (no GPU): layer schedule / filter design / padding vs the geometry
captured from the reference's full-width 256^2 generator, algorithmic work totals, synthetic inputs, module
construction and state-dict keys."""
import os

import numpy as np
import torch

from conftest import load_golden


def test_plan_matches_reference_layer_table():
    from afcm_amd import layer_schedule as sched
    g = load_golden('T256_layer_table')
    pl = sched.plan(256, 4, 1, {})
    layers = pl['enc'] + pl['dec']
    assert [L['name'] for L in layers] == [str(n) for n in g['names']]
    for L, row in zip(layers, g['table']):
        cin, cout, insz, outsz, up, down, ut, dt, p0, p1, p2, p3, k = [int(v) for v in row]
        assert (L['cin'], L['cout'], L['in_size'], L['out_size'], L['up'], L['down'], L['k']) == (cin, cout, insz, outsz, up, down, k)
        assert L['padding'] == [p0, p1, p2, p3]
        for key, f, taps in (('fu/', L['fu'], ut), ('fd/', L['fd'], dt)):
            assert (1 if f is None else len(f)) == taps
            if f is not None:
                assert np.abs(f.numpy() - g[key + L['name']]).max() <= 1e-7


def test_algorithmic_work_matches_baseline_md():
    """1085 MB fp32 / 622 MB 16-bit of filtered_lrelu traffic and 515 GFLOP of conv per image at 256^2."""
    from afcm_amd import layer_schedule as sched
    pl = sched.plan(256, 4, 1, {})
    w32 = sched.algorithmic_work(pl, 1, 4)
    w16 = sched.algorithmic_work(pl, 1, 2)
    assert abs(w32['filtered_lrelu_bytes'] / 1e6 - 1085.0) < 1.0
    assert abs(w16['filtered_lrelu_bytes'] / 1e6 - 622.0) < 1.0
    assert abs(w32['conv_flops'] / 1e9 - 515.0) < 1.0


def test_generator_module_keys_and_param_count():
    from afcm_amd.layer_schedule import DEFAULT_SYNTHESIS_KWARGS
    from afcm_amd.networks_stylegan3 import Stylegan3Generator
    g = load_golden('T256_layer_table')
    G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1,
                           mapping_kwargs=dict(num_layers=8), synthesis_kwargs=dict(DEFAULT_SYNTHESIS_KWARGS))
    assert list(G.state_dict().keys()) == [str(k) for k in g['sd_keys']]
    assert sum(p.numel() for p in G.parameters()) == int(g['nparams'])
    L = G.synthesis.L3_52_512
    assert (L.up_factor, L.down_factor, L.padding) == (4, 2, [-6, -9, -6, -9])
    # no CPU compute path: the forward must refuse CPU tensors loudly
    import pytest
    with pytest.raises(RuntimeError, match='no CPU'):
        G.mapping(torch.randn(1, 512), torch.rand(1, 1))


def test_synthetic_inputs():
    from afcm_amd import synthetic
    a, b, z, c = synthetic.generator_inputs(3, size=64, seed=1, slice_thickness=5)
    assert a.shape == (3, 4, 64, 64) and b.shape == (3, 1, 64, 64) and z.shape == (3, 512) and c.shape == (3, 1)
    assert a.min().item() == -1.0 and a.max().item() <= 1.0
    q = (a + 1) * 255 / 2
    assert torch.allclose(q, q.round(), atol=1e-4)                   # uint8-quantised like the reference's data
    assert set((c * 5).round().flatten().tolist()) <= {0.0, 1.0, 2.0, 3.0, 4.0}
    a2, *_ = synthetic.generator_inputs(3, size=64, seed=1, slice_thickness=5)
    assert torch.equal(a, a2)
    assert synthetic.psnr(b, b) > 100


def test_psnr_matches_oracle_definition():
    from afcm_amd import synthetic
    from oracle import aten_ops as ops
    torch.manual_seed(0)
    a = torch.rand(2, 1, 32, 32) * 2 - 1
    b = (a + 0.05 * torch.randn_like(a)).clamp(-1, 1)
    assert abs(synthetic.psnr(a, b) - ops.psnr(a, b)) < 1e-3


def test_scaled_linear_matches_the_reference_dense_layer_arithmetic():
    """FullyConnectedLayer's GEMM-alpha form (afcm_amd.networks_stylegan3._ScaledLinear) against the reference's
    x @ (w * weight_gain).t() + b * bias_gain (NET:97-100), first- and second-order gradients."""
    import torch
    from afcm_amd.networks_stylegan3 import _ScaledLinear
    torch.manual_seed(0)
    x = torch.randn(5, 7, dtype=torch.float64, requires_grad=True)
    w = torch.randn(3, 7, dtype=torch.float64, requires_grad=True)
    b = torch.randn(3, dtype=torch.float64, requires_grad=True)
    y = _ScaledLinear.apply(x, w, b, 0.3, 0.01)
    assert (y - (x @ (w * 0.3).t() + b * 0.01)).abs().max().item() < 1e-14
    assert torch.autograd.gradcheck(lambda x, w, b: _ScaledLinear.apply(x, w, b, 0.3, 0.01), (x, w, b))
    assert torch.autograd.gradgradcheck(lambda x, w, b: _ScaledLinear.apply(x, w, b, 0.3, 0.01), (x, w, b))
    assert torch.autograd.gradcheck(lambda x, w: _ScaledLinear.apply(x, w, None, 0.3, 1.0), (x, w))


def test_row_pitch_helpers():
    """afcm_amd/torch_utils/ops/_rows.py on CPU tensors: the pitched view, its recognition, and everything else falling back to dense."""
    import torch
    from afcm_amd.torch_utils.ops import _rows
    # rows to the next 64 bytes, unless that pads by more than 10 %
    assert _rows.pitch_for(276, torch.bfloat16) == 288 and _rows.pitch_for(278, torch.bfloat16) == 288 and _rows.pitch_for(148, torch.float16) == 160
    assert _rows.pitch_for(256, torch.float16) == 256 and _rows.pitch_for(84, torch.bfloat16) == 84 and _rows.pitch_for(36, torch.bfloat16) == 36
    t = _rows.empty([2, 3, 5, 276], torch.bfloat16, 'cpu')
    if _rows.ENABLED:
        assert tuple(t.shape) == (2, 3, 5, 276) and t.stride() == (3 * 5 * 288, 5 * 288, 288, 1) and _rows.pitch_of(t) == 288
        assert tuple(_rows.whole_buffer(t).shape) == (2, 3, 5, 288)
        assert _rows.rows(t)[0] is t
    assert _rows.empty([2, 3, 5, 276], torch.float32, 'cpu').is_contiguous()          # 16-bit streams only
    assert _rows.empty([2, 3, 5, 256], torch.bfloat16, 'cpu').is_contiguous()         # already whole lines
    d = torch.zeros(2, 3, 5, 276)
    assert _rows.pitch_of(d) == 276 and _rows.whole_buffer(d) is None
    v = torch.zeros(2, 3, 5, 300)[..., 4:280]                                          # some other view: made contiguous
    assert _rows.pitch_of(v[:, :, ::2]) is None
    r, ld = _rows.rows(v[:, :, ::2])
    assert r.is_contiguous() and ld == 276


def test_bench_gpus_n_launches_itself_and_the_ranks_report_missing_devices():
    """`python bench.py --gpus 2` with no launcher starts torch.distributed.run as a child (the parent makes no GPU
    call); on a host without two devices the RANKS say so and the exit code is non-zero -- no SystemExit in the parent."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['CUDA_VISIBLE_DEVICES'] = ''
    env['HIP_VISIBLE_DEVICES'] = ''
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert 'torch.distributed.run' in r.stderr
    assert 'rank 0: --gpus 2 needs 2 devices' in r.stderr and 'rank 1: --gpus 2 needs 2 devices' in r.stderr
