"""Functional CPU restatement of the conditional StyleGAN3 generator (oracle; test-only).

The network is evaluated from a plain ``state_dict`` (same keys as the reference's
``Stylegan3Generator.state_dict()``) plus a small config dict, using only ``oracle.aten_ops``.
NET = /root/reference/models/networks/stylegan3/networks_stylegan3.py.
"""
import math

import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F

from . import aten_ops as ops

DEFAULT_SYNTHESIS = dict(  # models/stylegan3_model.py:45-65
    channel_base=16384, channel_max=512, num_layers=14, num_critical=2, first_cutoff=2, first_stopband=2 ** 2.1,
    last_stopband_rel=2 ** 0.3, margin_size=10, output_scale=0.25, skip_resolution=128, conv_kernel=3, filter_size=6,
    lrelu_upsampling=2, conv_clamp=256, cond_mod=True)


def lowpass(numtaps, cutoff, width, fs):
    """Kaiser low-pass as the layers design it (NET:382-392); a single tap means 'no filter'."""
    if numtaps == 1:
        return None
    return torch.as_tensor(scipy.signal.firwin(numtaps=numtaps, cutoff=cutoff, width=width, fs=fs), dtype=torch.float32)


def _resample_geometry(in_size, out_size, in_sr, out_sr, in_cut, out_cut, in_hw, out_hw, k, filter_size, lrelu_up, torgb):
    """Up/down factors, tap counts, filters and padding of one layer (NET:294-334 / NET:453-489)."""
    tmp_sr = max(in_sr, out_sr) * (1 if torgb else lrelu_up)
    up = int(np.rint(tmp_sr / in_sr))
    down = int(np.rint(tmp_sr / out_sr))
    up_taps = filter_size * up if (up > 1 and not torgb) else 1
    down_taps = filter_size * down if (down > 1 and not torgb) else 1
    fu = lowpass(up_taps, in_cut, in_hw * 2, tmp_sr)
    fd = lowpass(down_taps, out_cut, out_hw * 2, tmp_sr)
    total = (out_size - 1) * down + 1 - (in_size + k - 1) * up + up_taps + down_taps - 2
    lo = (total + up) // 2
    hi = total - lo
    return dict(up=up, down=down, fu=fu, fd=fd, padding=[lo, hi, lo, hi])


def plan(img_resolution, img_channels_in, img_channels_out, synthesis_kwargs=None):
    """Layer schedule of SynthesisNetwork.__init__ (NET:589-664) as a list of plain dicts."""
    kw = dict(DEFAULT_SYNTHESIS)
    kw.update(synthesis_kwargs or {})
    n_layers, n_crit, margin = kw['num_layers'], kw['num_critical'], kw['margin_size']
    last_cut = img_resolution / 2
    last_stop = last_cut * kw['last_stopband_rel']
    e = np.minimum(np.arange(n_layers + 1) / (n_layers - n_crit), 1)
    cut = kw['first_cutoff'] * (last_cut / kw['first_cutoff']) ** e
    stop = kw['first_stopband'] * (last_stop / kw['first_stopband']) ** e
    sr = np.exp2(np.ceil(np.log2(np.minimum(stop * 2, img_resolution))))
    hw = np.maximum(stop, sr / 2) - cut
    sizes_enc = sr + margin * 2
    sizes = sizes_enc.copy()
    sizes[-2:] = img_resolution
    ch = np.rint(np.minimum((kw['channel_base'] / 2) / cut, kw['channel_max']))
    ch[-1] = img_channels_out

    k = kw['conv_kernel']
    enc = []
    for idx in range(n_layers):
        r = n_layers - idx - 1
        rp = n_layers - max(idx - 1, 0) - 1
        cin = img_channels_in if idx == 0 else int(ch[rp])
        g = _resample_geometry(int(sizes_enc[rp]), int(sizes_enc[r]), int(sr[rp]), int(sr[r]), cut[rp], cut[r], hw[rp], hw[r],
                               k, kw['filter_size'], kw['lrelu_upsampling'], False)
        g.update(name=f'encoder_{idx}', cin=cin, cout=int(ch[r]), in_size=int(sizes_enc[rp]), out_size=int(sizes_enc[r]), k=k,
                 store=bool(sizes[r] != sizes[rp] and sizes[rp] != sizes[0]), store_key=float(sizes[r]))
        enc.append(g)

    skip_res = kw['skip_resolution']
    res_log2 = int(np.log2(img_resolution))
    if skip_res >= 4:
        fs = int(np.log2(skip_res))
        skip_connects = [True] * (fs - 1) + [False] * (res_log2 - fs)
    else:
        skip_connects = [False] * res_log2

    dec = []
    res_idx = 1
    for idx in range(n_layers + 1):
        p = max(idx - 1, 0)
        torgb = idx == n_layers
        kk = 1 if torgb else k
        g = _resample_geometry(int(sizes[p]), int(sizes[idx]), int(sr[p]), int(sr[idx]), cut[p], cut[idx], hw[p], hw[idx],
                               kk, kw['filter_size'], kw['lrelu_upsampling'], torgb)
        nxt = min(idx + 1, n_layers)
        if sizes[idx] != sizes[nxt] and sizes[idx] != sizes[0]:          # NET:693-697
            skip = skip_connects[res_idx]
            res_idx += 1
        else:
            skip = False
        g.update(name=f'L{idx}_{int(sizes[idx])}_{int(ch[idx])}', cin=int(ch[p]), cout=int(ch[idx]), in_size=int(sizes[p]),
                 out_size=int(sizes[idx]), k=kk, torgb=torgb, skip=bool(skip), skip_key=float(sizes[idx]))
        dec.append(g)
    return dict(enc=enc, dec=dec, margin=margin, output_scale=kw['output_scale'], conv_clamp=kw['conv_clamp'],
                cond_mod=kw['cond_mod'], c0=int(ch[0]), num_ws=n_layers + 2, kw=kw)


def fully_connected(x, weight, bias, act='linear', lr_mul=1.0):
    """FullyConnectedLayer.forward (NET:89-101)."""
    w = weight * (lr_mul / math.sqrt(weight.shape[1]))
    b = bias * lr_mul if bias is not None else None
    if act == 'linear' and b is not None:
        return torch.addmm(b[None], x, w.t())
    return ops.bias_act(x @ w.t(), b, act=act)


def mapping(sd, z, c, num_ws, num_layers, lr_mul=0.01, prefix='mapping.'):
    """MappingNetwork.forward with truncation_psi=1, update_emas=False (NET:135-161)."""
    dt = sd[prefix + 'fc0.weight'].dtype          # fp32 normally; fp64 when the oracle is run as ground truth
    x = z.to(dt)
    x = x * (x.square().mean(1, keepdim=True) + 1e-8).rsqrt()
    if (prefix + 'embed.weight') in sd:
        y = fully_connected(c.to(dt), sd[prefix + 'embed.weight'], sd[prefix + 'embed.bias'])
        y = y * (y.square().mean(1, keepdim=True) + 1e-8).rsqrt()
        x = torch.cat([x, y], 1)
    for i in range(num_layers):
        x = fully_connected(x, sd[f'{prefix}fc{i}.weight'], sd[f'{prefix}fc{i}.bias'], act='lrelu', lr_mul=lr_mul)
    return x[:, None].repeat(1, num_ws, 1)


def _instr(name, codes, preact):
    kw = {}
    if codes is not None and name in codes:
        kw['codes'] = codes[name]
    if preact is not None:
        kw['record'] = preact.setdefault(name, [])
    return kw


def synthesis(sd, pl, ws, img_in, dropout_mask=None, prefix='synthesis.', taps=None, codes=None, preact=None):
    """SynthesisNetwork.forward (NET:666-705) with update_emas=False.

    ``dropout_mask`` (already scaled by 1/(1-p)) stands in for torch's Dropout in training mode;
    ``None`` is eval mode.  ``taps`` optionally receives every resampling layer's output; ``preact`` (dict) every layer's
    pre-activation tensor and ``codes`` (dict name -> uint8 codes) imposes another implementation's leaky-ReLU branch decisions
    (see oracle.aten_ops.filtered_lrelu).
    """
    dt = sd[prefix + 'fc_in.weight'].dtype
    ws = ws.to(dt).unbind(1)
    m = pl['margin']
    x = F.pad(img_in.to(dt), [m] * 4)
    feats = {}
    clamp = pl['conv_clamp']
    for L in pl['enc']:                                                    # EncoderLayer.forward NET:491-516
        p = prefix + L['name'] + '.'
        w = sd[p + 'weight'] * (1 / math.sqrt(L['cin'] * L['k'] ** 2))
        x = ops.conv2d(x, w, padding=L['k'] - 1)
        x = ops.filtered_lrelu(x, fu=L['fu'], fd=L['fd'], b=sd[p + 'bias'], up=L['up'], down=L['down'], padding=L['padding'],
                               gain=math.sqrt(2), slope=0.2, clamp=clamp, **_instr(L['name'], codes, preact))
        if taps is not None:
            taps[L['name']] = x
        if L['store']:
            feats[L['store_key']] = x

    # bottleneck (NET:682-686): Conv2dLayer 3x3 lrelu (CoModGAN/layers.py:153-162) -> 4x4 avg-pool -> FC lrelu -> dropout
    p = prefix + 'e_16x16.'
    w = sd[p + 'weight']
    g = ops.conv2d(x, w * (1 / math.sqrt(w.shape[1] * w.shape[2] ** 2)), padding=w.shape[2] // 2)
    g = ops.bias_act(g, sd[p + 'bias'], act='lrelu')
    g = F.adaptive_avg_pool2d(g, (4, 4)).flatten(1)
    g = fully_connected(g, sd[prefix + 'fc_in.weight'], sd[prefix + 'fc_in.bias'], act='lrelu')
    if dropout_mask is not None:
        g = g * dropout_mask

    for L, w_lat in zip(pl['dec'], ws[1:]):                                # SynthesisLayer.forward NET:336-379
        p = prefix + L['name'] + '.'
        lat = torch.cat([w_lat, g], 1) if pl['cond_mod'] else w_lat
        styles = fully_connected(lat, sd[p + 'affine.weight'], sd[p + 'affine.bias'])
        if L['torgb']:
            styles = styles * (1 / math.sqrt(L['cin'] * L['k'] ** 2))
        input_gain = sd[p + 'magnitude_ema'].rsqrt()
        x = ops.modulated_conv2d(x, sd[p + 'weight'], styles, demodulate=not L['torgb'], padding=L['k'] - 1,
                                 input_gain=input_gain)
        x = ops.filtered_lrelu(x, fu=L['fu'], fd=L['fd'], b=sd[p + 'bias'], up=L['up'], down=L['down'], padding=L['padding'],
                               gain=1.0 if L['torgb'] else math.sqrt(2), slope=1.0 if L['torgb'] else 0.2, clamp=clamp,
                               **_instr(L['name'], codes, preact))
        if L['skip']:
            x = x + feats[L['skip_key']]
        if taps is not None:
            taps[L['name']] = x
    if pl['output_scale'] != 1:
        x = x * pl['output_scale']
    return x.to(dt)


def generator(sd, pl, z, c, cond_img, mapping_layers, dropout_mask=None, taps=None, codes=None, preact=None):
    """Stylegan3Generator.forward (NET:737-740)."""
    ws = mapping(sd, z, c, pl['num_ws'], mapping_layers)
    return synthesis(sd, pl, ws, cond_img, dropout_mask=dropout_mask, taps=taps, codes=codes, preact=preact)


def random_state_dict(pl, z_dim, c_dim, w_dim, mapping_layers, seed=0, lr_mul=0.01):
    """Random-init parameters with the reference's shapes and init statistics (NET:83-85,308-311,463-466)."""
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gen)
    sd = {}
    for L in pl['enc']:
        p = 'synthesis.' + L['name'] + '.'
        sd[p + 'weight'] = rn(L['cout'], L['cin'], L['k'], L['k'])
        sd[p + 'bias'] = torch.zeros(L['cout'])
        sd[p + 'magnitude_ema'] = torch.ones([])
    c0 = pl['c0']
    sd['synthesis.e_16x16.weight'] = rn(c0, c0, 3, 3)
    sd['synthesis.e_16x16.bias'] = torch.zeros(c0)
    sd['synthesis.fc_in.weight'] = rn(1024, c0 * 16)
    sd['synthesis.fc_in.bias'] = torch.zeros(1024)
    gdim = 1024 if pl['cond_mod'] else 0
    for L in pl['dec']:
        p = 'synthesis.' + L['name'] + '.'
        sd[p + 'weight'] = rn(L['cout'], L['cin'], L['k'], L['k'])
        sd[p + 'bias'] = torch.zeros(L['cout'])
        sd[p + 'magnitude_ema'] = torch.ones([])
        sd[p + 'affine.weight'] = rn(L['cin'], w_dim + gdim)
        sd[p + 'affine.bias'] = torch.ones(L['cin'])
    sd['mapping.w_avg'] = torch.zeros(w_dim)
    if c_dim > 0:
        sd['mapping.embed.weight'] = rn(w_dim, c_dim)
        sd['mapping.embed.bias'] = torch.zeros(w_dim)
    feats = [z_dim + (w_dim if c_dim > 0 else 0)] + [w_dim] * mapping_layers
    for i in range(mapping_layers):
        sd[f'mapping.fc{i}.weight'] = rn(feats[i + 1], feats[i]) / lr_mul
        sd[f'mapping.fc{i}.bias'] = torch.zeros(feats[i + 1])
    return sd
