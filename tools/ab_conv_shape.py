#!/usr/bin/env python3
"""A/B of the MFMA shape of the 16-bit 3x3 conv kernel: conv2d_fwd16_kernel (v_mfma_f32_32x32x16, K-chunks of 16
channels) against conv2d_fwd16x_kernel (v_mfma_f32_16x16x32, chunks of 32) at the same workgroup and wave tile, and conv2d_wgrad16g_kernel in
its two shapes (X16 template flag), per generator layer: forward, data gradient and weight gradient, on RANDOM data, both kernels in ONE library and ONE process, rounds interleaved (cdna_hip_programming.md rule 24).
Needs a library built with -DAFCM_CONV_AB (both kernels + afcm_debug_conv_variant):
    tools/build_variant.sh conv_ab conv2d.hip "-DAFCM_CONV_AB"
    AFCM_HIP_LIB=$PWD/afcm_amd/csrc/variants/conv_ab.so python tools/ab_conv_shape.py [--rounds 7]
With a -DAFCM_CONV_AB -DAFCM_CONV_STAMPS build (--stamps) it prints instead, per layer and shape, the median K-loop cycles per workgroup
and the shader clock held inside the K loop (d s_memtime / d s_memrealtime x 100 MHz, stamped around the loop; MI355X_MICROARCH.md
'DVFS give-back' item 6)."""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import _lib
from afcm_amd import layer_schedule as sched
from afcm_amd.torch_utils.ops import conv2d as C

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16); ap.add_argument('--rounds', type=int, default=7); ap.add_argument('--iters', type=int, default=4)
ap.add_argument('--stamps', action='store_true'); ap.add_argument('--zeros', action='store_true')
ap.add_argument('--variants', default='0,1', help='two values of afcm_debug_conv_variant to compare (bit 0: 16x16x32, bit 1: row blocks fastest)')
a = ap.parse_args()
lib = _lib.load()
dbg = ctypes.CDLL(_lib.LIB_PATH)
assert hasattr(dbg, 'afcm_debug_conv_variant'), 'build conv2d.hip with -DAFCM_CONV_AB (see the docstring)'
dt = torch.bfloat16
VA, VB = [int(v) for v in a.variants.split(',')]
NAMES = {0: '32x32x16', 1: '16x16x32', 3: '16x16x32/ob', 33: '16x16x32/2wg', 49: '16x16x32/3wg', 65: '16x16x32/4wg', 97: '16x16x32/6wg'}
VMAP = {0: VA, 1: VB}


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def stamps(fn):
    nb = 65536
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    assert dbg.afcm_debug_conv_stamps_clear() == 0
    fn(); torch.cuda.synchronize()
    cyc = np.zeros([nb, 4], dtype=np.uint64); rt = np.zeros([nb, 4], dtype=np.uint64)
    assert dbg.afcm_debug_conv_stamps(cyc.ctypes.data_as(ctypes.c_void_p), nb) == 0
    assert dbg.afcm_debug_conv_realtime(rt.ctypes.data_as(ctypes.c_void_p), nb) == 0
    ok = (cyc[:, 3] > cyc[:, 0]) & (cyc[:, 0] > 0)
    c, r = cyc[ok].astype(np.int64), rt[ok].astype(np.int64)
    kl, kr = c[:, 2] - c[:, 1], r[:, 2] - r[:, 1]
    good = kr > 0
    extra = ''
    # realtime (100 MHz) view of the launch: its span, and how many stamped units (workgroups; tiles of the persistent kernel) were alive on average
    span_rt = int(r[:, 3].max() - r[:, 0].min())
    alive = float((r[:, 3] - r[:, 0]).sum()) / max(1, span_rt)
    extra = f' [{ok.sum()} units, span {span_rt / 100.0:.1f} us, {alive:.0f} alive on average]'
    if hasattr(dbg, 'afcm_debug_conv_prologue'):
        pro = np.zeros([nb, 4], dtype=np.uint64)
        assert dbg.afcm_debug_conv_prologue(pro.ctypes.data_as(ctypes.c_void_p), nb) == 0
        pr = pro[ok].astype(np.int64)
        if (pr[:, 0] > 0).any():          # (the 16x16x32 kernel stamps its prologue: entry -> requests issued -> all returned -> patch written -> barrier)
            extra += ' [prologue: ' + ' / '.join(f'{np.median(v):.0f}' for v in (pr[:, 0] - c[:, 0], pr[:, 1] - pr[:, 0], pr[:, 2] - pr[:, 1], c[:, 1] - pr[:, 2])) + ']'
    return float(np.median(kl)), float(np.median(kl[good] / kr[good]) * 0.1), float(np.median(c[:, 1] - c[:, 0])), float(np.median(c[:, 3] - c[:, 2])), extra    # K loop cycles, GHz, prologue, epilogue


pl = sched.plan(256, 4, 1, {})
seen = set()
KINDS = ('fwd', 'dgrad', 'wgrad')
tot = {(v, k): 0.0 for v in (0, 1) for k in KINDS}
flops_tot = 0.0
print(f'# batch {a.batch}, bf16, {"zeros" if a.zeros else "random"} operands; ' + ('median shader cycles per workgroup: prologue + K loop + epilogue, clock held inside the K loop'
      if a.stamps else f'ms per launch: median of {a.rounds} interleaved rounds x {a.iters} launches; TF/s = algorithmic flops / that'))
for L in pl['enc'] + pl['dec']:
    n, ci, co, h, k = a.batch, L['cin'], L['cout'], L['in_size'], L['k']
    if k != 3:
        continue
    pad = k - 1
    x = torch.randn(n, ci, h, h, device='cuda', dtype=dt)
    w = torch.randn(co, ci, k, k, device='cuda')
    if a.zeros:
        x.zero_(); w.zero_()
    packs, ys = {}, {}
    for v in (0, 1):
        assert dbg.afcm_debug_conv_variant(VMAP[v]) == 0
        packs[v] = (C.pack_weights(w, dt, 0), C.pack_weights(w, dt, 1))
        ys[v] = C._conv_raw(x, packs[v][0][0], packs[v][0][1], None, co, k, pad)
    err = (ys[0].float() - ys[1].float()).abs().max().item() / max(1e-9, ys[0].float().abs().max().item())
    y = ys[0]
    fl = 2.0 * n * co * ci * k * k * y.shape[2] * y.shape[3]

    def run(v, kind):
        dbg.afcm_debug_conv_variant(VMAP[v])
        (wp, rp), (wpt, rpt) = packs[v]
        if kind == 'fwd':
            return lambda: C._conv_raw(x, wp, rp, None, co, k, pad)
        if kind == 'wgrad':
            return lambda: C._wgrad_raw(y, x, co, ci, k, pad)
        return lambda: C._conv_raw(y, wpt, rpt, None, ci, k, k - 1 - pad)

    key = (ci, co, h)
    if a.stamps:
        if key in seen:
            continue
        seen.add(key)
        out = []
        for kind in ('fwd', 'dgrad'):
            for v in (0, 1):
                dbg.afcm_debug_conv_variant(VMAP[v])
                cyc, ghz, pro, epi, extra = stamps(run(v, kind))
                out.append(f'{kind} {NAMES[VMAP[v]]} {pro:6.0f} + {cyc:7.0f} + {epi:6.0f} cyc {ghz:4.2f} GHz{extra}')
        print(f'{L["name"]:14s} {ci:3d}->{co:3d} @{h:3d}  ' + ' | '.join(out))
        continue
    dws = {}
    for v in (0, 1):
        dws[v] = run(v, 'wgrad')()
    werr = (dws[0] - dws[1]).abs().max().item() / max(1e-9, dws[0].abs().max().item())
    res = {}
    for kind in KINDS:
        for v in (0, 1):
            run(v, kind)(); run(v, kind)()
        samples = {0: [], 1: []}
        for r in range(a.rounds):
            for v in ((0, 1) if r % 2 == 0 else (1, 0)):
                dbg.afcm_debug_conv_variant(VMAP[v])
                samples[v].append(timed(run(v, kind), a.iters))
        for v in (0, 1):
            res[(v, kind)] = float(np.median(samples[v]))
            tot[(v, kind)] += res[(v, kind)]
    flops_tot += fl
    if key not in seen:
        seen.add(key)
        print(f'{L["name"]:14s} {ci:3d}->{co:3d} @{h:3d}  ' + ' | '.join(
            f'{kind} {NAMES[VA]} {res[(0, kind)]:6.3f} ms {fl / res[(0, kind)] / 1e9:6.0f} TF  {NAMES[VB]} {res[(1, kind)]:6.3f} ms {fl / res[(1, kind)] / 1e9:6.0f} TF  x{res[(0, kind)] / res[(1, kind)]:.3f}'
            for kind in KINDS) + f'  (outputs differ by {err:.1e}, weight gradients by {werr:.1e} of scale)')
if not a.stamps:
    print('TOTAL (29 layers) ' + ' | '.join(f'{kind} {NAMES[VA]} {tot[(0, kind)]:.2f} ms {flops_tot / tot[(0, kind)] / 1e9:.0f} TF  {NAMES[VB]} {tot[(1, kind)]:.2f} ms '
                                           f'{flops_tot / tot[(1, kind)] / 1e9:.0f} TF  x{tot[(0, kind)] / tot[(1, kind)]:.3f}' for kind in KINDS))
