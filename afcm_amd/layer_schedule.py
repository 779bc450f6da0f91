"""Host-side layer geometry of the conditional StyleGAN3 generator: band-limits, sampling rates, sizes,
channel counts, Kaiser filters and filtered_lrelu paddings.

Pure numpy/scipy host logic shared by the network modules (networks_stylegan3.py), the benchmarks and
the byte/flop accounting; no tensors on any device.  Follows the reference's constructor arithmetic:
schedule NET:589-611, per-layer factors/filters/padding NET:294-334 (decoder) and NET:453-489 (encoder),
filter design NET:382-392.  (NET = models/networks/stylegan3/networks_stylegan3.py in the reference.)
"""
import numpy as np
import scipy.signal
import scipy.special
import torch

# models/stylegan3_model.py:45-65 -- the shipped `--model stylegan3` defaults
DEFAULT_SYNTHESIS_KWARGS = dict(
    channel_base=16384, channel_max=512, num_layers=14, num_critical=2, first_cutoff=2, first_stopband=2 ** 2.1,
    last_stopband_rel=2 ** 0.3, margin_size=10, output_scale=0.25, skip_resolution=128, conv_kernel=3, filter_size=6,
    lrelu_upsampling=2, use_radial_filters=False, conv_clamp=256, magnitude_ema_beta=0.5 ** (16 / 20e3), cond_mod=True)


def design_lowpass_filter(numtaps, cutoff, width, fs, radial=False):
    """Kaiser-windowed low-pass (separable) or jinc-based radial filter; one tap means identity (None)."""
    assert numtaps >= 1
    if numtaps == 1:
        return None
    if not radial:
        return torch.as_tensor(scipy.signal.firwin(numtaps=numtaps, cutoff=cutoff, width=width, fs=fs), dtype=torch.float32)
    t = (np.arange(numtaps) - (numtaps - 1) / 2) / fs
    r = np.hypot(*np.meshgrid(t, t))
    f = scipy.special.j1(2 * cutoff * (np.pi * r)) / (np.pi * r)
    beta = scipy.signal.kaiser_beta(scipy.signal.kaiser_atten(numtaps, width / (fs / 2)))
    w = np.kaiser(numtaps, beta)
    f = f * np.outer(w, w)
    return torch.as_tensor(f / np.sum(f), dtype=torch.float32)


def resample_geometry(in_size, out_size, in_sampling_rate, out_sampling_rate, conv_kernel, filter_size, lrelu_upsampling,
                      is_torgb=False):
    """(tmp_sampling_rate, up_factor, up_taps, down_factor, down_taps, padding[4]) of one layer."""
    in_size = np.broadcast_to(np.asarray(in_size), [2])
    out_size = np.broadcast_to(np.asarray(out_size), [2])
    tmp_sr = max(in_sampling_rate, out_sampling_rate) * (1 if is_torgb else lrelu_upsampling)
    up = int(np.rint(tmp_sr / in_sampling_rate))
    assert in_sampling_rate * up == tmp_sr
    down = int(np.rint(tmp_sr / out_sampling_rate))
    assert out_sampling_rate * down == tmp_sr
    up_taps = filter_size * up if up > 1 and not is_torgb else 1
    down_taps = filter_size * down if down > 1 and not is_torgb else 1
    pad_total = (out_size - 1) * down + 1            # desired size before downsampling
    pad_total = pad_total - (in_size + conv_kernel - 1) * up   # size after upsampling
    pad_total = pad_total + up_taps + down_taps - 2  # shrink caused by the two 'valid' FIRs
    pad_lo = (pad_total + up) // 2                   # symmetric sample-location convention
    pad_hi = pad_total - pad_lo
    padding = [int(pad_lo[0]), int(pad_hi[0]), int(pad_lo[1]), int(pad_hi[1])]
    return tmp_sr, up, up_taps, down, down_taps, padding


def band_schedule(img_resolution, img_channels_out, num_layers, num_critical, first_cutoff, first_stopband, last_stopband_rel,
                  margin_size, channel_base, channel_max):
    """Geometric progression of cutoffs/stopbands and everything derived from it (NET:596-611)."""
    last_cutoff = img_resolution / 2
    last_stopband = last_cutoff * last_stopband_rel
    exponents = np.minimum(np.arange(num_layers + 1) / (num_layers - num_critical), 1)
    cutoffs = first_cutoff * (last_cutoff / first_cutoff) ** exponents
    stopbands = first_stopband * (last_stopband / first_stopband) ** exponents
    sampling_rates = np.exp2(np.ceil(np.log2(np.minimum(stopbands * 2, img_resolution))))
    half_widths = np.maximum(stopbands, sampling_rates / 2) - cutoffs
    sizes_for_encoder = sampling_rates + margin_size * 2
    sizes = sizes_for_encoder.copy()
    sizes[-2:] = img_resolution
    channels = np.rint(np.minimum((channel_base / 2) / cutoffs, channel_max))
    channels[-1] = img_channels_out
    return dict(cutoffs=cutoffs, stopbands=stopbands, sampling_rates=sampling_rates, half_widths=half_widths,
                sizes=sizes, sizes_for_encoder=sizes_for_encoder, channels=channels)


def plan(img_resolution=256, img_channels_in=4, img_channels_out=1, synthesis_kwargs=None):
    """Flat description of every resampling layer (14 encoder + 15 decoder at the defaults): channels, sizes,
    up/down factors, filters, padding.  Used by benchmarks and the byte/flop accounting."""
    kw = dict(DEFAULT_SYNTHESIS_KWARGS)
    kw.update(synthesis_kwargs or {})
    n = kw['num_layers']
    bs = band_schedule(img_resolution, img_channels_out, n, kw['num_critical'], kw['first_cutoff'], kw['first_stopband'],
                       kw['last_stopband_rel'], kw['margin_size'], kw['channel_base'], kw['channel_max'])
    cut, hw, sr, sizes, sizes_e, ch = (bs[k] for k in ('cutoffs', 'half_widths', 'sampling_rates', 'sizes', 'sizes_for_encoder', 'channels'))
    enc, dec = [], []
    for idx in range(n):
        r = n - idx - 1
        rp = n - max(idx - 1, 0) - 1
        tmp_sr, up, ut, down, dt, pad = resample_geometry(int(sizes_e[rp]), int(sizes_e[r]), int(sr[rp]), int(sr[r]),
                                                           kw['conv_kernel'], kw['filter_size'], kw['lrelu_upsampling'])
        enc.append(dict(name=f'encoder_{idx}', cin=img_channels_in if idx == 0 else int(ch[rp]), cout=int(ch[r]),
                        in_size=int(sizes_e[rp]), out_size=int(sizes_e[r]), k=kw['conv_kernel'], up=up, down=down, padding=pad,
                        fu=design_lowpass_filter(ut, cut[rp], hw[rp] * 2, tmp_sr), fd=design_lowpass_filter(dt, cut[r], hw[r] * 2, tmp_sr),
                        torgb=False, modulated=False))
    for idx in range(n + 1):
        p = max(idx - 1, 0)
        torgb = idx == n
        k = 1 if torgb else kw['conv_kernel']
        tmp_sr, up, ut, down, dt, pad = resample_geometry(int(sizes[p]), int(sizes[idx]), int(sr[p]), int(sr[idx]), k,
                                                           kw['filter_size'], kw['lrelu_upsampling'], torgb)
        dec.append(dict(name=f'L{idx}_{int(sizes[idx])}_{int(ch[idx])}', cin=int(ch[p]), cout=int(ch[idx]), in_size=int(sizes[p]),
                        out_size=int(sizes[idx]), k=k, up=up, down=down, padding=pad,
                        fu=design_lowpass_filter(ut, cut[p], hw[p] * 2, tmp_sr), fd=design_lowpass_filter(dt, cut[idx], hw[idx] * 2, tmp_sr),
                        torgb=torgb, modulated=True))
    return dict(enc=enc, dec=dec, schedule=bs, kw=kw)


def algorithmic_work(pl, batch, elem_size):
    """Per-step algorithmic work of the generator forward (SURVEY.md section 8d / BASELINE.md section 3).

    filtered_lrelu bytes: read the conv output once, write the layer output once, write 2 bits per element
    of the sign grid.  Conv flops: 2 * Cout * Cin * k^2 * (in + k - 1)^2 per image.
    """
    fl_bytes = 0
    conv_flops = 0
    for L in pl['enc'] + pl['dec']:
        h = L['in_size'] + L['k'] - 1
        fuw = 1 if L['fu'] is None else len(L['fu'])
        fdw = 1 if L['fd'] is None else len(L['fd'])
        sh = L['out_size'] * L['down'] - (L['down'] - 1) + fdw - 1
        swb = ((sh + 15) & ~15) // 4
        fl_bytes += batch * L['cout'] * (elem_size * (h * h + L['out_size'] ** 2) + sh * swb)
        conv_flops += batch * 2 * L['cout'] * L['cin'] * L['k'] ** 2 * h * h
        del fuw
    return dict(filtered_lrelu_bytes=fl_bytes, conv_flops=conv_flops)
