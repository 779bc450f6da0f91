"""Row f3 host logic (CPU): image metrics and the sliding-window predictor.

Predictor arithmetic is pinned to vectors captured from the reference's own functions (tools/gen_golden_predictor.py);
PSNR / SSIM restate scikit-image (absent here) and are pinned to closed forms and brute-force window loops only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from afcm_amd import evaluation as ev
from afcm_amd import predictor as pr


def test_gen_indices_and_remove_halo_match_reference_vectors():
    g = load_golden('P1_predictor')
    for n, (i, k, s) in enumerate(g['gi_args']):
        vol = (int(i), int(k), int(k))
        got = sorted({ix[0].start for ix in pr.patch_indices(vol, (int(k), int(k), int(k)), (int(s), int(k), int(k)))})
        assert got == sorted(set(g[f'gi_{n}'].tolist()))
    shape = tuple(int(v) for v in g['rh_shape'])
    for n, m in enumerate(g['rh_meta']):
        halo, (z0, z1, y0, y1, x0, x1), want_idx = tuple(m[:3]), m[3:9], m[9:]
        patch = g[f'rh_patch_{n}']
        got, idx = pr.remove_halo(patch, (slice(0, 2), slice(z0, z1), slice(y0, y1), slice(x0, x1)), shape, tuple(int(h) for h in halo))
        assert np.array_equal(got, g[f'rh_out_{n}']), n
        assert [v for s in idx[1:] for v in (s.start, s.stop)] == want_idx.tolist(), n


def test_sliding_window_identity_model_reconstructs_the_volume():
    rng = np.random.default_rng(1)
    vol = rng.standard_normal((1, 12, 40, 44)).astype(np.float32)
    p = pr.SlidingWindowPredictor(out_channels=1, patch_halo=(2, 4, 4))
    out = p.run(lambda b: b, vol, patch_shape=(8, 16, 16), stride_shape=(4, 8, 8), batch_size=3)
    assert out.shape == vol.shape and np.allclose(out, vol, atol=1e-6)
    # a model with a border artefact inside the halo: the halo removal hides it everywhere but on the volume border
    def edgy(b):
        b = b.clone()
        b[..., 0, :, :] += 5; b[..., :, 0, :] += 5; b[..., :, :, 0] += 5
        return b
    out = p.run(edgy, vol, patch_shape=(8, 16, 16), stride_shape=(4, 8, 8))
    assert np.allclose(out[:, 1:, 1:, 1:], vol[:, 1:, 1:, 1:], atol=1e-6)
    with pytest.raises(AssertionError):
        pr.validate_halo((4, 8, 8), (8, 16, 16), (6, 16, 16))


def _ssim_brute(x, y, win=7, R=2.0):
    pad = win // 2
    c1, c2 = (0.01 * R) ** 2, (0.03 * R) ** 2
    vals = []
    for i in range(pad, x.shape[0] - pad):
        for j in range(pad, x.shape[1] - pad):
            a = x[i - pad:i + pad + 1, j - pad:j + pad + 1].astype(np.float64).ravel()
            b = y[i - pad:i + pad + 1, j - pad:j + pad + 1].astype(np.float64).ravel()
            ua, ub = a.mean(), b.mean()
            va, vb = a.var(ddof=1), b.var(ddof=1)
            vab = ((a - ua) * (b - ub)).sum() / (a.size - 1)
            vals.append(((2 * ua * ub + c1) * (2 * vab + c2)) / ((ua ** 2 + ub ** 2 + c1) * (va + vb + c2)))
    return float(np.mean(vals))


def test_ssim_matches_window_loops_and_psnr_closed_form():
    rng = np.random.default_rng(2)
    x = rng.random((19, 23)).astype(np.float32)
    y = np.clip(x + 0.1 * rng.standard_normal(x.shape).astype(np.float32), 0, 1)
    assert abs(ev.structural_similarity(x, y) - _ssim_brute(x, y)) < 1e-10
    assert abs(ev.structural_similarity(x, x) - 1.0) < 1e-12
    assert abs(ev.structural_similarity(x, y) - ev.structural_similarity(y, x)) < 1e-12
    v = rng.random((9, 10, 11))
    assert 0 < ev.structural_similarity(v, np.clip(v + 0.05, 0, 1)) < 1            # 3-D window
    mse = np.mean((x.astype(np.float64) - y) ** 2)
    assert abs(ev.peak_signal_noise_ratio(x, y) - 10 * np.log10(1.0 / mse)) < 1e-10          # non-negative reference: range 1
    xs, ys = x - 0.5, y - 0.5
    mse_s = np.mean((xs.astype(np.float64) - ys) ** 2)
    assert abs(ev.peak_signal_noise_ratio(xs, ys) - 10 * np.log10(4.0 / mse_s)) < 1e-10     # signed reference: range 2
    assert abs(ev.peak_signal_noise_ratio(x, y, data_range=0.5) - 10 * np.log10(0.25 / mse)) < 1e-10
    with pytest.raises(ValueError):
        ev.peak_signal_noise_ratio(x * 3, y)
    assert ev.peak_signal_noise_ratio(x, x) == float('inf')


def test_evaluate_2d_and_volume_metrics_follow_the_reference_bookkeeping():
    rng = np.random.default_rng(3)
    L = rng.random((4, 1, 1, 16, 16)).astype(np.float32)
    G = np.clip(L + 0.05 * rng.standard_normal(L.shape).astype(np.float32), 0, 1)
    L[2] = 0                                             # an empty target slice is skipped
    psnr, ssim, mae = ev.evaluate_2D(G, L)
    keep = [0, 1, 3]
    want_psnr = np.mean([ev.peak_signal_noise_ratio(L[i, 0, 0] / L[i, 0, 0].max(), G[i, 0, 0] / G[i, 0, 0].max()) for i in keep])
    assert abs(psnr - want_psnr) < 1e-9
    assert abs(ssim - np.mean([ev.structural_similarity(L[i, 0, 0], G[i, 0, 0]) for i in keep])) < 1e-12
    assert abs(mae - np.mean(np.abs(L - G))) < 1e-7      # whole-batch MAE, as the reference computes it
    assert ev.evaluate_2D(G, np.zeros_like(L)) is None
    Lv, Gv = rng.random((8, 9, 10)), None
    Gv = np.clip(Lv + 0.02 * rng.standard_normal(Lv.shape), 0, 1)
    p3, s3, m3 = ev.evaluate_one(Gv, Lv)
    assert 20 < p3 < 60 and 0 < s3 <= 1 and abs(m3 - np.mean(np.abs(Lv - Gv))) < 1e-12
    ps, ss, _ = ev.evaluate_slice(Gv, Lv)
    assert 20 < ps < 60 and 0 < ss <= 1
    pv, sv, _ = ev.evaluate_3D(Gv, Lv)
    assert abs(pv - ev.peak_signal_noise_ratio(Lv, Gv)) < 1e-12 and 0 < sv <= 1
    assert np.array_equal(ev.to_unit_range(np.array([-3.0, -1.0, 0.0, 1.0, 2.0])), np.array([0, 0, 0.5, 1, 1], dtype=np.float32))


def test_checkpoint_names_round_trip_and_ema_bookkeeping(tmp_path):
    """Row f4 (checkpoint half): '<epoch>_net_<name>.pth' files with bare-module keys (models/base_model.py:144-199), loadable
    back -- also from a 'module.'-prefixed (DataParallel) state-dict -- and the EMA update of train.py:67-77."""
    from afcm_amd.stylegan3_model import StyleGAN3GeneratorStep, update_ema

    class Tiny(torch.nn.Module):
        z_dim, c_dim = 4, 1

        def __init__(self):
            super().__init__()
            self.mapping = torch.nn.Linear(4, 4)
            self.synthesis = torch.nn.Linear(4, 4)
            self.register_buffer('w_avg', torch.zeros(4))

    torch.manual_seed(0)
    net = Tiny()
    step = StyleGAN3GeneratorStep.__new__(StyleGAN3GeneratorStep)       # host-side bookkeeping only: no optimizer kernel on CPU
    step.netG, step.netG_ema, step.model_names = net, __import__('copy').deepcopy(net).eval(), ['G', 'G_ema']
    with torch.no_grad():
        for p in net.parameters():
            p.add_(1.0)
        net.w_avg.fill_(3.0)
    beta = step.update_ema(batch_size=16, total_iters=160, ema_kimgs=10.0, ramp=0.05)
    assert abs(beta - 0.5 ** (16 / min(10000.0, 160 * 0.05))) < 1e-12
    for pe, p in zip(step.netG_ema.parameters(), net.parameters()):
        assert torch.allclose(pe, p.detach() - 1.0 + (1 - beta) * 1.0, atol=1e-6)       # p_ema <- p.lerp(p_ema, beta)
    assert torch.equal(step.netG_ema.w_avg, net.w_avg)                                  # buffers are copied
    step.save_networks('latest', str(tmp_path))
    assert sorted(f.name for f in tmp_path.iterdir()) == ['latest_net_G.pth', 'latest_net_G_ema.pth']
    sd = torch.load(tmp_path / 'latest_net_G.pth', weights_only=True)
    assert list(sd.keys()) == list(net.state_dict().keys())
    fresh = StyleGAN3GeneratorStep.__new__(StyleGAN3GeneratorStep)
    fresh.netG, fresh.netG_ema, fresh.model_names = Tiny(), Tiny(), ['G', 'G_ema']
    fresh.load_networks('latest', str(tmp_path))
    for a, b in zip(fresh.netG.state_dict().values(), net.state_dict().values()):
        assert torch.equal(a, b)
    torch.save({'module.' + k: v for k, v in net.state_dict().items()}, tmp_path / 'dp_net_G.pth')
    torch.save(step.netG_ema.state_dict(), tmp_path / 'dp_net_G_ema.pth')
    fresh.load_networks('dp', str(tmp_path))
    assert torch.equal(fresh.netG.mapping.weight, net.mapping.weight)
    assert update_ema(step.netG_ema, net, 16, 10 ** 9, ema_kimgs=10.0, ramp=None) == 0.5 ** (16 / 10000.0)


def _volumes(d=23, h=40, w=36, seed=3):
    rng = np.random.default_rng(seed)
    return {'flair': rng.integers(0, 256, (d, h, w)).astype(np.uint8), 't1_hr4sr': rng.integers(0, 256, (d, h, w)).astype(np.uint8)}


def test_slice_dataset_thick_slices_and_fraction():
    """afcm_amd.data.SliceDataset, the shipped loader configuration (slice_num 4, thickness [5], one modality in, one out; data/
    cmsr_dataset.py:98-155): A = the thick slices at -1, 0, +1, +2 thicknesses around the target's own, zero planes (-> -1 after
    normalisation) outside the volume; B = the target slice; slice_idx = the offset inside the thick slice / thickness."""
    from afcm_amd import data
    vols = _volumes()
    ds = data.SliceDataset(vols, phase='val', patch_shape=(1, 32, 32), stride_shape=(1, 32, 32), raw_internal_path_in=['flair'],
                           raw_internal_path_out=['t1_hr4sr'], thickness=[5], slice_num=4)
    assert len(ds) == 23                                                    # one patch per slice: the crop made every slice 32 x 32
    crop = lambda v: v[:, 4:36, 2:34].astype(np.float64)                     # centre crop 40 x 36 -> 32 x 32
    norm = lambda m: np.clip(2 * (m / 255.0) - 1, -1, 1).astype(np.float32)
    fl, t1 = crop(vols['flair']), crop(vols['t1_hr4sr'])
    for idx in (0, 3, 7, 14, 19, 22):
        it = ds[idx]
        base = (idx // 5) * 5
        assert it['A'].shape == (4, 32, 32) and it['B'].shape == (1, 32, 32) and it['A'].dtype == torch.float32
        for k, pos in enumerate((base - 5, base, base + 5, base + 10)):
            want = norm(fl[pos]) if 0 <= pos <= 22 else np.full((32, 32), -1.0, dtype=np.float32)
            assert np.array_equal(it['A'][k].numpy(), want), (idx, k)
        assert np.array_equal(it['B'][0].numpy(), norm(t1[idx]))
        assert np.allclose(it['slice_idx'], [(idx - base) / 5]) and it['slice_idx'].dtype == np.float32
        assert it['B_idx'].item() == idx and it['B_class'].tolist() == [1.0]
    with pytest.raises(StopIteration):
        ds[23]
    # test phase: (A, fraction, position); slice_num 1: the slice itself
    a, frac, where = data.SliceDataset(vols, phase='test', patch_shape=(1, 32, 32), stride_shape=(1, 32, 32), raw_internal_path_in=['flair'],
                                       raw_internal_path_out=['t1_hr4sr'], thickness=[5], slice_num=4)[8]
    assert a.shape == (4, 32, 32) and np.allclose(frac.numpy(), [3 / 5]) and where[0] == slice(8, 9)
    one = data.SliceDataset(vols, phase='val', patch_shape=(1, 32, 32), stride_shape=(1, 32, 32), raw_internal_path_in=['flair'],
                            raw_internal_path_out=['t1_hr4sr'], thickness=[5], slice_num=1)[8]
    assert one['A'].shape == (1, 32, 32) and np.array_equal(one['A'][0].numpy(), norm(fl[8]))


def test_slice_dataset_crop_pad_normalise_and_chain():
    from afcm_amd import data
    v = np.arange(2 * 5 * 7, dtype=np.float64).reshape(2, 5, 7)
    got = data.crop_to_fixed(v, (9, 4))                                      # pad rows 5 -> 9 (2 above, 2 below), crop columns 7 -> 4 from 1
    assert got.shape == (2, 9, 4) and np.array_equal(got[:, 2:7], v[:, :, 1:5]) and not got[:, :2].any() and not got[:, 7:].any()
    assert np.array_equal(data.crop_to_fixed(v, (5, 7)), v)
    assert np.allclose(data.normalize(np.array([0.0, 127.5, 255.0, 300.0])), [-1, 0, 1, 1])
    # patches larger strides: spatial patches inside a slice, last one pulled back to the border (data/utils.py:117-122)
    ds = data.SliceDataset({'raw': np.zeros((3, 40, 40), np.uint8)}, phase='val', patch_shape=(1, 32, 32), stride_shape=(1, 32, 32), slice_num=1)
    assert len(ds) == 3
    ds = data.SliceDataset({'raw': np.zeros((3, 80, 80), np.uint8)}, phase='val', patch_shape=(1, 80, 80), stride_shape=(1, 32, 32), slice_num=1)
    assert len(ds) == 3 and ds[0]['A'].shape == (1, 80, 80)
    both = data.cmsr_dataset([_volumes(seed=1), _volumes(d=11, seed=2)], phase='val', patch_shape=(1, 32, 32), stride_shape=(1, 32, 32),
                             raw_internal_path_in=['flair'], raw_internal_path_out=['t1_hr4sr'], thickness=[5], slice_num=4)
    assert len(both) == 23 + 11
    batch = next(iter(torch.utils.data.DataLoader(both, batch_size=4)))
    assert batch['A'].shape == (4, 4, 32, 32) and batch['B'].shape == (4, 1, 32, 32) and batch['slice_idx'].shape == (4, 1)
    with pytest.raises(RuntimeError, match='h5py'):
        data.SliceDataset('/nonexistent/subject.h5', raw_internal_path_in=['flair'], raw_internal_path_out=['t1_hr4sr'])
