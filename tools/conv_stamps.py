#!/usr/bin/env python3
"""Where a conv workgroup's lifetime goes: run with AFCM_HIP_LIB=<a build with -DAFCM_CONV_STAMPS> (tools/build_variant.sh); prints, per shape, the
median shader cycles a workgroup spends in its prologue (entry -> first barrier), K loop and epilogue, and its share of the launch.
    AFCM_HIP_LIB=afcm_amd/csrc/variants/conv_stamps.so python tools/conv_stamps.py [cin cout size]..."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import _lib
from afcm_amd.torch_utils.ops import conv2d as C
lib = ctypes.CDLL(_lib.LIB_PATH)
args = [int(v) for v in sys.argv[1:]] or [64, 64, 276, 128, 128, 276, 512, 512, 84]
def nkc_of(ci):
    return (ci + 15) // 16


for ci, co, h in zip(args[0::3], args[1::3], args[2::3]):
    x = torch.randn(16, ci, h, h, device='cuda', dtype=torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, device='cuda')
    wp, rp = C.pack_weights(w, torch.bfloat16, 0)
    for _ in range(3):
        y = C._conv_raw(x, wp, rp, None, co, 3, 2)
    torch.cuda.synchronize()
    assert lib.afcm_debug_conv_stamps_clear() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = C._conv_raw(x, wp, rp, None, co, 3, 2); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nb = 65536
    buf = np.zeros([nb, 4], dtype=np.uint64)
    rc = lib.afcm_debug_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb)
    assert rc == 0
    ok = (buf[:, 3] > buf[:, 0]) & (buf[:, 0] > 0)
    if not ok.any():
        print('no stamps', buf[:4]); continue
    bar = np.zeros([nb, 4], dtype=np.uint64)
    assert lib.afcm_debug_conv_barrier_cycles(bar.ctypes.data_as(ctypes.c_void_p), nb) == 0
    bw = bar[ok].astype(np.int64)
    phb = np.zeros([nb, 4], dtype=np.uint64)
    assert lib.afcm_debug_conv_phase_cycles(phb.ctypes.data_as(ctypes.c_void_p), nb) == 0
    phw = phb[ok].astype(np.int64) / nkc_of(ci)
    b = buf[ok].astype(np.int64)
    pro, kl, epi = b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 3] - b[:, 2]
    span = b[:, 3].max() - b[:, 0].min()
    nkc = (ci + 15) // 16
    mi = 2 if co > 64 and ((co + 63) // 64 * 64) % 128 == 0 else 1
    print(f'{ci:3d}->{co:3d} @{h:3d}: {ms:.3f} ms, {ok.sum()} workgroups stamped, launch span {span} cycles ({span / ms / 1e6:.2f} GHz); '
          f'median cycles per workgroup: prologue {np.median(pro):.0f}, K loop {np.median(kl):.0f} ({np.median(kl) / nkc:.0f} per chunk; MFMA pipe alone '
          f'{mi * 4 * 9 * 32} per chunk and wave), epilogue {np.median(epi):.0f}; total {np.median(b[:, 3] - b[:, 0]):.0f}; of the K loop, cycles waiting at its barriers (median per wave 0..3): '
          + ' '.join(f'{np.median(bw[:, w]):.0f}' for w in range(4)) + f'; fastest / slowest wave of a workgroup (median): {np.median(bw.max(1)):.0f} / {np.median(bw.min(1)):.0f}'
          + '; wave 0, cycles per chunk: top..tap 2 / ..tap 4 / tap 5 + patch write / ..barrier: ' + ' '.join(f'{np.median(phw[:, i]):.0f}' for i in range(4)))
