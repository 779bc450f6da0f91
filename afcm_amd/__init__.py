"""afcm_amd -- MI355X (gfx950) native hot path of the AFCM `--model stylegan3` generator.

Layout
  csrc/                  hand-written HIP kernels + the C ABI (include/afcm_hip.h) -> libafcm_hip.so
  _lib.py                ctypes binding of the C ABI (fails loudly when the library is missing)
  torch_utils/ops/       host-side mirror of the reference's fused-op API
                         (filtered_lrelu / upfirdn2d / bias_act / conv2d_gradfix / modulated conv)
  networks_stylegan3.py  drop-in generator (same class names, forward() signatures, state-dict keys)
  stylegan3_model.py     the generator part of the training step (run_G, fwd+bwd)
  distributed.py         one-process-per-GPU data parallelism: bucketed RCCL all-reduce of gradients

There is no CPU fallback in this package: ops raise if the tensors are not on a ROCm device or the
HIP library is not built.  The CPU restatement used for parity checks lives in the top-level
``oracle/`` package and is test infrastructure only.
"""
__version__ = '0.1.0'

from .torch_utils import op_registry as _op_registry  # noqa: E402,F401  (registers torch.ops.afcm.*; loads no native code)
