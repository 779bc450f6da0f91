"""Fused gradient scrub + Adam for the generator update (C ABI ``afcm_adam_multi``).

The reference's G update is ``nan_to_num`` on every gradient followed by ``torch.optim.Adam(lr, betas=(0, 0.99))``
(models/stylegan3_model.py:122-124,132-135; models/comodgan_model.py:19-20).  Run eagerly that is ~110 tiny launches plus
seven multi-tensor passes over the 234 MB of parameters; here one HIP launch reads p, g, m, v and writes p, m, v once for
every tensor.  State layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter) and the update arithmetic are
torch.optim.Adam's, so ``state_dict()`` round-trips with it.
"""
import ctypes
import math

import torch

from . import _lib


class FusedScrubAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, scrub=True, posinf=1e5, neginf=-1e5, write_grad=False, capturable=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError('invalid Adam hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.scrub, self.posinf, self.neginf, self.write_grad = bool(scrub), float(posinf), float(neginf), bool(write_grad)
        self._tables = {}          # (group index, device) -> (rows, device table, pinned host copy)
        # capturable: the step count is a device scalar per group and the bias corrections are formed in the kernel (C ABI
        # afcm_adam_multi_capturable), so the launch can be captured into a hipGraph and replayed (bench.py --graph); state['step'] then
        # stays where it was at capture time -- `device_step()` is the count
        self.capturable = bool(capturable)
        self._step_dev = {}

    def _state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = torch.tensor(0.0, dtype=torch.float32)
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None, grads=None, grad_scale=1.0):
        """One update.  ``grads``: optional {parameter: gradient tensor} (e.g. slices of the all-reduce buckets, so the
        reduced values are consumed in place); default ``p.grad``.  ``grad_scale`` multiplies every gradient first
        (1 / world size after a sum all-reduce)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        chunk = lib.afcm_adam_chunk_elems()
        for gi, group in enumerate(self.param_groups):
            beta1, beta2 = group['betas']
            rows = []
            touched = []
            chunks = 0
            step_t = None
            for p in group['params']:
                g = grads.get(p) if grads is not None else p.grad
                if g is None:
                    continue
                if p.dtype != torch.float32 or g.dtype != torch.float32:
                    raise RuntimeError('FusedScrubAdam updates float32 parameters with float32 gradients')
                _lib.require_gpu(p, g)
                if not p.is_contiguous():
                    raise RuntimeError('FusedScrubAdam needs contiguous parameters')
                if not g.is_contiguous():
                    g = g.contiguous()
                    if grads is None:
                        p.grad = g
                st = self._state(p)
                if step_t is None:
                    # one step count per group (bias corrections are launch scalars): the largest per-parameter count + 1.
                    # torch.optim.Adam counts per parameter; the two only differ for a parameter that skipped a step.
                    step_t = int(max(float(self._state(q)['step']) for q in group['params'])) + 1
                st['step'].fill_(step_t)
                rows.append((p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel(), chunks))
                touched.extend((p, st['exp_avg'], st['exp_avg_sq']))
                chunks += (p.numel() + chunk - 1) // chunk
            if not rows:
                continue
            dev = group['params'][0].device
            key = (gi, dev)
            ent = self._tables.get(key)
            if ent is None or ent[0] != rows:
                # the pointer table only changes when the allocator hands out different gradient blocks: steady-state steps
                # skip the upload.  A fresh pinned tensor per change: an in-flight copy never sees it rewritten.
                if self.capturable:
                    # ONE pinned buffer for the optimizer's life (nothing is allocated while a stream capture is under way): rewritten in
                    # place -- after a synchronize in eager steps (an earlier upload may not have run yet), as it is during a capture (the
                    # captured copy node reads the buffer at every replay: the captured step is then the only one that may run)
                    new = torch.tensor(rows, dtype=torch.int64).reshape(-1)
                    if ent is None:
                        host = torch.empty(6 * len(group['params']), dtype=torch.int64).pin_memory()
                        table = torch.empty(6 * len(group['params']), dtype=torch.int64, device=dev)
                    else:
                        table, host = ent[1], ent[2]
                    if not torch.cuda.is_current_stream_capturing():
                        torch.cuda.current_stream(dev).synchronize()
                    host[:new.numel()].copy_(new)
                    table[:new.numel()].copy_(host[:new.numel()], non_blocking=True)
                    self._tables[key] = ent = (rows, table, host)
                else:
                    host = torch.tensor(rows, dtype=torch.int64).reshape(-1).pin_memory()
                    table = ent[1] if (ent is not None and ent[1].numel() >= host.numel()) else torch.empty(6 * len(group['params']), dtype=torch.int64, device=dev)
                    table[:host.numel()].copy_(host, non_blocking=True)
                    self._tables[key] = ent = (rows, table, host)
            table = ent[1]
            if self.capturable:
                sd = self._step_dev.get(key)
                if sd is None:
                    sd = self._step_dev[key] = torch.full([1], float(step_t - 1), dtype=torch.float32, device=dev)
                _lib.check(lib.afcm_adam_multi_capturable(ctypes.c_void_p(table.data_ptr()), len(rows), chunks, ctypes.c_void_p(sd.data_ptr()), group['lr'],
                                                          beta1, beta2, group['eps'], float(grad_scale), int(self.scrub), self.posinf, self.neginf,
                                                          int(self.write_grad), _lib.stream_ptr(table)), 'adam_multi_capturable')
                torch.autograd.graph.increment_version(touched)
                continue
            # bias corrections in double on the host, as torch.optim.Adam does for non-capturable steps
            bc1 = 1.0 - beta1 ** step_t
            bc2 = 1.0 - beta2 ** step_t
            _lib.check(lib.afcm_adam_multi(ctypes.c_void_p(table.data_ptr()), len(rows), chunks, group['lr'] / bc1, beta1, beta2,
                                           1.0 - beta1, 1.0 - beta2, math.sqrt(bc2), group['eps'], float(grad_scale), int(self.scrub), self.posinf, self.neginf,
                                           int(self.write_grad), _lib.stream_ptr(table)), 'adam_multi')
            # the kernel wrote through raw pointers: tell autograd (version counters), so that anything keyed on a parameter's version --
            # saved-tensor checks, caches of derived tensors such as packed weight images -- sees the update
            torch.autograd.graph.increment_version(touched)
        return loss


    def device_step(self, group=0):
        """The step count of a capturable optimizer (a device scalar; reading it synchronises)."""
        for (gi, _), t in self._step_dev.items():
            if gi == group:
                return int(t.item())
        return 0


class _WeightedL1(torch.autograd.Function):
    """weight * mean(|a - b|) for fp32 device tensors in two launches forward (partials + their sum) and one backward (C ABI afcm_l1_partials /
    afcm_l1_grad); gradient for `a` only."""

    @staticmethod
    def forward(ctx, a, b, weight):
        _lib.require_gpu(a, b)
        lib = _lib.load()
        a, b = a.contiguous(), b.contiguous()
        numel = a.numel()
        blocks = max(1, min(256, (numel + 4095) // 4096))
        partials = torch.empty([blocks], dtype=torch.float32, device=a.device)
        _lib.check(lib.afcm_l1_partials(partials.data_ptr(), a.data_ptr(), b.data_ptr(), numel, blocks, float(weight), _lib.stream_ptr(a)), 'l1_partials')
        ctx.save_for_backward(a, b)
        ctx.weight = float(weight)
        return partials.sum()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        gout = gout.to(torch.float32).contiguous()
        _lib.check(_lib.load().afcm_l1_grad(ga.data_ptr(), a.data_ptr(), b.data_ptr(), gout.data_ptr(), a.numel(), ctx.weight, _lib.stream_ptr(a)), 'l1_grad')
        return ga, None, None


def weighted_l1(a, b, weight):
    """``torch.nn.L1Loss()(a, b) * weight`` (models/stylegan3_model.py:107); the fused form for fp32 device tensors of one shape where only `a`
    needs a gradient, the op-by-op composition otherwise."""
    if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape and not b.requires_grad and a.numel() > 0:
        return _WeightedL1.apply(a, b, float(weight))
    return torch.nn.functional.l1_loss(a, b) * weight
