#!/usr/bin/env python3
"""In-loop shader clock of the filtered_lrelu wave kernels, in the training step and stand-alone (VERDICT r05 #2d).

    tools/build_variant.sh wavestamps filtered_lrelu_wave.hip -DAFCM_WAVE_STAMPS        (here or on the GPU box)
    AFCM_HIP_LIB=$PWD/afcm_amd/csrc/variants/wavestamps.so python tools/flrelu_clock.py

The diagnostic build stamps s_memtime / s_memrealtime around every strip and sums the deltas per kernel variant (up, down, sign mode, strip
height).  Two phases on one box, each after >= 2 s of the same work: (a) bench.py's training step (bf16, batch 16), (b) the same kernels
launched back to back on the layer's own tensors with nothing else on the chip (forward + transposed op of every generator layer).
clock = cycles / ticks x 100 MHz; strip time = ticks / strips x 10 ns."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from afcm_amd import _lib, layer_schedule as sched, synthetic
from afcm_amd.networks_stylegan3 import Stylegan3Generator
from afcm_amd.stylegan3_model import StyleGAN3GeneratorStep

lib = _lib.load()
for fn in ('afcm_debug_wave_stamps', 'afcm_debug_wave_stamps_clear'):
    if not hasattr(lib, fn):
        raise SystemExit('this library has no stamps: build it with -DAFCM_WAVE_STAMPS and select it with AFCM_HIP_LIB')
lib.afcm_debug_wave_stamps.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 128)()


def read():
    torch.cuda.synchronize()
    assert lib.afcm_debug_wave_stamps(buf) == 0
    out = {}
    for slot in range(32):
        cyc, tick, strips = buf[4 * slot], buf[4 * slot + 1], buf[4 * slot + 2]
        if strips:
            name = f'up{4 if slot & 1 else 2} down{4 if slot & 2 else 2} {("none", "write", "read", "read-aligned")[(slot >> 2) & 3]} {"48" if slot & 16 else "32"}-row'
            out[name] = (cyc / tick * 100.0, tick / strips * 0.01, strips)
    return out


dev = torch.device('cuda', 0)
torch.manual_seed(0)
G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1, mapping_kwargs=dict(num_layers=8),
                       synthesis_kwargs=dict(sched.DEFAULT_SYNTHESIS_KWARGS, compute_dtype=torch.bfloat16)).to(dev).train()
step = StyleGAN3GeneratorStep(G, lr_G=0.0025, lambda_L1=100.0)
inputs = synthetic.generator_inputs(16, size=256, seed=0, device=dev)


def one():
    step.set_input(*inputs)
    step.optimize_parameters()


t0 = time.time()
while time.time() - t0 < 2.5:
    one()
torch.cuda.synchronize()
lib.afcm_debug_wave_stamps_clear()
for _ in range(20):
    one()
in_step = read()

# stand-alone: every layer's forward + transposed op, back to back, nothing else on the chip
from afcm_amd.torch_utils.ops import filtered_lrelu as flr
cases = []
S = G.synthesis
for layer in [getattr(S, f'encoder_{i}') for i in range(S.num_layers)] + [getattr(S, n) for n in S.layer_names]:
    if layer.up_filter is None or layer.down_filter is None or layer.up_filter.numel() < 2:
        continue                     # (the 1x1 ToRGB layer has identity filters: the pointwise kernel)
    x = torch.randn(16, layer.out_channels, int(layer.in_size[0]) + 2, int(layer.in_size[0]) + 2, device=dev).to(torch.bfloat16).requires_grad_(True)
    cases.append((x, dict(fu=layer.up_filter, fd=layer.down_filter, up=layer.up_factor, down=layer.down_factor, padding=layer.padding, gain=2 ** 0.5,
                          slope=0.2, clamp=256.0)))


def alone():
    for x, kw in cases:
        y = flr.filtered_lrelu(x, **kw)
        y.backward(torch.ones_like(y))
        x.grad = None


t0 = time.time()
while time.time() - t0 < 2.5:
    alone()
torch.cuda.synchronize()
lib.afcm_debug_wave_stamps_clear()
for _ in range(10):
    alone()
stand = read()
print('# filtered_lrelu wave kernels: in-loop shader clock (GHz) and time per strip (us), in the training step | stand-alone')
for k in sorted(set(in_step) | set(stand)):
    a, b = in_step.get(k), stand.get(k)
    fa = f'{a[0] / 1000:5.3f} GHz {a[1]:7.2f} us ({a[2]} strips)' if a else '-'
    fb = f'{b[0] / 1000:5.3f} GHz {b[1]:7.2f} us ({b[2]} strips)' if b else '-'
    print(f'{k:34s} step {fa:42s} alone {fb}')
