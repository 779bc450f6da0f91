#!/usr/bin/env python3
"""Per-layer filtered_lrelu micro-benchmark (GPU): forward (sign write) and backward (sign read) of every
resampling layer of the 256^2 generator at batch B, reported as algorithmic GB/s (SURVEY.md 8d:
read x + write y + 2 bits/elem of signs)."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afcm_amd import layer_schedule as sched  # noqa: E402
from afcm_amd.torch_utils.ops import filtered_lrelu as flr  # noqa: E402


def raw_layer(L, x, fu, fd, kw, args):
    """Forward (sign write) and transposed backward (sign read) through filtered_lrelu._run on dense or row-pitched tensors."""
    from afcm_amd.torch_utils.ops import _rows, fused_layer
    pitched = args.raw == 'pitched'
    cfg = fused_layer._cfg(kw['up'], kw['down'], kw['padding'], kw['gain'], kw['slope'], kw['clamp'])
    if pitched:
        xp = _rows.empty(list(x.shape), x.dtype, x.device)
        xp.copy_(x)
        x = xp
    y, signs, layout, _ = flr._run(x, fu, fd, None, None, cfg, True, pitched_out=pitched)
    g = _rows.empty(list(y.shape), y.dtype, y.device, pitched=pitched)
    g.copy_(torch.randn(y.shape, device=y.device, dtype=y.dtype))
    bcfg = flr._backward_cfg(cfg, fu, fd, x.shape, y.shape, layout)
    # --rotate K: K copies of every operand, taken in turn (and the results kept alive in a ring of K), so that no launch finds its operands
    # in the 256 MB Infinity Cache or in L2 because the previous launch of the same layer left them there
    K = max(1, args.rotate)
    def copy_of(t):
        c = _rows.empty(list(t.shape), t.dtype, t.device, pitched=not t.is_contiguous())
        c.copy_(t)
        return c
    xs, gs, ss = [x] + [copy_of(x) for _ in range(K - 1)], [g] + [copy_of(g) for _ in range(K - 1)], [signs] + [signs.clone() for _ in range(K - 1)]
    keep = [None] * K
    tf = tb = float('inf')
    for _rep in range(args.repeats):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for i in range(2):
            keep[i % K] = flr._run(xs[i % K], fu, fd, None, None, cfg, True, pitched_out=pitched)
        ev[0].record()
        for i in range(args.iters):
            keep[i % K] = flr._run(xs[i % K], fu, fd, None, None, cfg, True, pitched_out=pitched)
        ev[1].record()
        for i in range(2):
            keep[i % K] = flr._run(gs[i % K], fd, fu, None, ss[i % K], bcfg, False, pitched_out=pitched)
        ev[2].record()
        for i in range(args.iters):
            keep[i % K] = flr._run(gs[i % K], fd, fu, None, ss[i % K], bcfg, False, pitched_out=pitched)
        ev[3].record()
        torch.cuda.synchronize()
        tf = min(tf, ev[0].elapsed_time(ev[1]) / args.iters)
        tb = min(tb, ev[2].elapsed_time(ev[3]) / args.iters)
    nbytes = (x.numel() + y.numel()) * x.element_size() + signs.numel()
    return tf, tb, nbytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--repeats', type=int, default=3, help='timed repeats per layer; the fastest is reported (box noise is +-10 %)')
    ap.add_argument('--layers', default='', help='comma-separated layer names (default: all)')
    ap.add_argument('--rotate', type=int, default=1, help='raw mode: cycle through this many copies of the operands (cold caches)')
    ap.add_argument('--no-bias', action='store_true', help='b=None: the generator path (the convs add the bias)')
    ap.add_argument('--raw', choices=['dense', 'pitched'], default=None,
                    help='drive the launches as the fused layer node does (filtered_lrelu._run, no autograd), on dense or row-pitched tensors')
    args = ap.parse_args()
    dt = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}[args.dtype]
    pl = sched.plan(args.res, 4, 1, {})
    tot_b = tot_t = tot_bb = tot_tb = 0.0
    seen = {}
    only = set(args.layers.split(',')) if args.layers else None
    for L in pl['enc'] + pl['dec']:
        if only is not None and L['name'] not in only:
            continue
        h = L['in_size'] + L['k'] - 1
        key = (L['cout'], h, L['up'], L['down'], tuple(L['padding']))
        x = torch.randn(args.batch, L['cout'], h, h, device='cuda', dtype=dt).requires_grad_(True)
        b = None if args.no_bias else torch.zeros(L['cout'], device='cuda', dtype=dt)
        fu = None if L['fu'] is None else L['fu'].cuda()
        fd = None if L['fd'] is None else L['fd'].cuda()
        kw = dict(up=L['up'], down=L['down'], padding=L['padding'], gain=1.0 if L.get('torgb') else math.sqrt(2),
                  slope=1.0 if L.get('torgb') else 0.2, clamp=256.0)
        if args.raw:
            tf, tb, nbytes = raw_layer(L, x.detach(), fu, fd, kw, args)
            tot_b += nbytes; tot_t += tf; tot_bb += nbytes; tot_tb += tb
            if key not in seen:
                seen[key] = 1
                print(f'{L["name"]:14s} C={L["cout"]:3d} {h:3d}->{L["out_size"]:3d} up{L["up"]} down{L["down"]}  fwd {tf:7.3f} ms {nbytes/tf/1e6:7.1f} GB/s   bwd {tb:7.3f} ms {nbytes/tb/1e6:7.1f} GB/s')
            continue
        y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=b, **kw)
        signs = y.grad_fn.saved_tensors[2]
        r = torch.randn_like(y)
        tf = tb = float('inf')
        for _rep in range(args.repeats):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for _ in range(2):
                y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=b, **kw)
            ev[0].record()
            for _ in range(args.iters):
                y = flr.filtered_lrelu(x, fu=fu, fd=fd, b=b, **kw)
            ev[1].record()
            for _ in range(2):
                torch.autograd.grad(y, x, r, retain_graph=True)
            ev[2].record()
            for _ in range(args.iters):
                torch.autograd.grad(y, x, r, retain_graph=True)
            ev[3].record()
            torch.cuda.synchronize()
            tf = min(tf, ev[0].elapsed_time(ev[1]) / args.iters)
            tb = min(tb, ev[2].elapsed_time(ev[3]) / args.iters)
        nbytes = (x.numel() + y.numel()) * x.element_size() + signs.numel()
        tot_b += nbytes; tot_t += tf; tot_bb += nbytes; tot_tb += tb
        if key not in seen:
            seen[key] = 1
            print(f'{L["name"]:14s} C={L["cout"]:3d} {h:3d}->{L["out_size"]:3d} up{L["up"]} down{L["down"]}  fwd {tf:7.3f} ms {nbytes/tf/1e6:7.1f} GB/s   bwd {tb:7.3f} ms {nbytes/tb/1e6:7.1f} GB/s')
        del x, y, r, signs
    print(f'TOTAL fwd {tot_t:.2f} ms = {tot_b/tot_t/1e6:.1f} GB/s ({tot_t/args.batch:.3f} ms/img); bwd {tot_tb:.2f} ms = {tot_bb/tot_tb/1e6:.1f} GB/s; bytes/img {tot_b/args.batch/1e6:.1f} MB')


if __name__ == '__main__':
    main()
