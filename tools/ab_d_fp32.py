"""A/B of the discriminator's fp32 blocks (the 16^2 and smaller resolutions): framework convolution (MIOpen) against the MFMA route
(3x3: split bf16 operands).  usage: python tools/ab_d_fp32.py    (GPU box, repo root) -- runs bench.py --with-discriminator twice."""
import json
import subprocess
import sys

for flag in ('0', '1'):
    code = ("import sys; import afcm_amd.torch_utils.ops.conv2d_resample as r; r.MFMA_CONV_FP32 = bool(%s); import bench; "
            "sys.argv = ['bench.py', '--with-discriminator', '--steps', '6', '--cpu-baseline', 'off']; bench.main()" % flag)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if not line:
        print('MFMA_CONV_FP32 =', flag, 'failed', out.stderr[-600:])
        continue
    d = json.loads(line[-1])
    print('MFMA_CONV_FP32 =', flag, round(d['ms_per_step'], 2), 'ms/iteration', round(d['value'], 1), d['unit'])
