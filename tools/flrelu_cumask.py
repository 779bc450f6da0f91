#!/usr/bin/env python3
"""Is a filtered_lrelu launch bound inside the CU or by what the CUs share (L2 / fabric / HBM)?

Runs the forward (sign-writing) and transposed (sign-reading) launch of a few generator layers on streams restricted to a subset of the
chip's CUs (hipExtStreamCreateWithCUMask).  A CU-local bound (issue slots, dependency chains, LDS) gives time ~ 1 / CUs; a shared bound
gives a per-CU rate that rises as CUs are taken away.  Evidence tool (GPU box): python tools/flrelu_cumask.py [--layers a,b]"""
import argparse
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import layer_schedule as sched  # noqa: E402
from afcm_amd.torch_utils.ops import filtered_lrelu as flr  # noqa: E402
from afcm_amd.torch_utils.ops import _rows, fused_layer  # noqa: E402


def masked_stream(hip, words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', default='encoder_1,encoder_2,encoder_5,encoder_8')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    hip = ctypes.CDLL('libamdhip64.so')
    torch.cuda.init()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    nw = (ncu + 31) // 32
    masks = {
        'all': [0xffffffff] * nw,
        'half_alternate': [0x55555555] * nw,
        'half_low_words': [0xffffffff] * (nw // 2) + [0] * (nw - nw // 2),
        'half_even_words': [0xffffffff if i % 2 == 0 else 0 for i in range(nw)],
        'quarter_alternate': [0x11111111] * nw,
    }
    streams = {k: masked_stream(hip, v) for k, v in masks.items()}
    pl = sched.plan(256, 4, 1, {})
    only = set(args.layers.split(','))
    print(f'{ncu} CUs; time per launch in ms, (relative to all CUs)')
    for L in pl['enc'] + pl['dec']:
        if L['name'] not in only:
            continue
        h = L['in_size'] + L['k'] - 1
        x0 = torch.randn(args.batch, L['cout'], h, h, device='cuda', dtype=torch.bfloat16)
        fu, fd = L['fu'].cuda(), L['fd'].cuda()
        cfg = fused_layer._cfg(L['up'], L['down'], L['padding'], math.sqrt(2), 0.2, 256.0)
        x = _rows.empty(list(x0.shape), x0.dtype, x0.device)
        x.copy_(x0)
        y, signs, layout, _ = flr._run(x, fu, fd, None, None, cfg, True, pitched_out=True)
        g = _rows.empty(list(y.shape), y.dtype, y.device, pitched=True)
        g.copy_(torch.randn(y.shape, device='cuda', dtype=y.dtype))
        bcfg = flr._backward_cfg(cfg, fu, fd, x.shape, y.shape, layout)
        torch.cuda.synchronize()
        base = None
        for name, st in streams.items():
            with torch.cuda.stream(st):
                res = []
                for fwd in (True, False):
                    best = float('inf')
                    for _rep in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        for i in range(args.iters + 2):
                            if i == 2:
                                e0.record()
                            if fwd:
                                flr._run(x, fu, fd, None, None, cfg, True, pitched_out=True)
                            else:
                                flr._run(g, fd, fu, None, signs, bcfg, False, pitched_out=True)
                        e1.record()
                        st.synchronize()
                        best = min(best, e0.elapsed_time(e1) / args.iters)
                    res.append(best)
            if base is None:
                base = res
            print(f'{L["name"]:12s} C={L["cout"]:3d} {h:3d}^2  {name:18s} fwd {res[0]:7.3f} ({res[0] / base[0]:4.2f}x)   bwd {res[1]:7.3f} ({res[1] / base[1]:4.2f}x)')


if __name__ == '__main__':
    main()
