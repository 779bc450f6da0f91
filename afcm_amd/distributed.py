"""One-process-per-GPU data parallelism for the generator: bucketed all-reduce of gradients over RCCL
(torch.distributed backend "nccl" on ROCm; "gloo" in CPU tests), launched from autograd hooks so the
collectives overlap the rest of backward.

The reference trains G on a single device (its DataParallel wrapper is bypassed for G,
models/comodgan_model.py:14-15); sharding the batch over ranks is this build's
multi-GPU story.  The path shards by batch with ONE exchange step per iteration: sum-all-reduce of G's
gradients (58.5 M params = 234 MB fp32), averaged over ranks.

Buckets are filled in reverse parameter order (the order backward produces gradients), ~25 MB each:
on xGMI (7 point-to-point links per GPU) a few large ring collectives beat many small ones, and the
last-layer buckets are in flight while the encoder's backward still runs.
"""
import torch
import torch.distributed as dist


class GradientBuckets:
    def __init__(self, params, bucket_bytes=25 * 1024 * 1024, process_group=None, comm_dtype=None, force=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())   # force: exercise the hooks/collectives on one rank
        self.params = [p for p in params if p.requires_grad]
        self.comm_dtype = comm_dtype
        self._buckets = []          # list of dict(params, flat, pending, handle)
        self._where = {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._add_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._add_bucket(cur)
        self._hooks = []
        if self.active:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._inflight = []

    def _add_bucket(self, plist):
        idx = len(self._buckets)
        total = sum(p.numel() for p in plist)
        dev = plist[0].device
        dtype = self.comm_dtype or plist[0].dtype
        self._buckets.append(dict(params=list(plist), flat=torch.zeros(total, dtype=dtype, device=dev), pending=len(plist), n=len(plist)))
        off = 0
        for p in plist:
            self._where[p] = (idx, off)
            off += p.numel()

    @property
    def num_buckets(self):
        return len(self._buckets)

    def broadcast_parameters(self, module, src=0):
        """Initial weight (and buffer) broadcast from rank `src` so that every replica starts identical."""
        if not self.active:
            return
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)

    def _on_grad(self, p):
        idx, off = self._where[p]
        b = self._buckets[idx]
        b['flat'][off:off + p.numel()].copy_(p.grad.reshape(-1))
        b['pending'] -= 1
        if b['pending'] == 0:
            # every gradient of this bucket is final: launch its all-reduce now, while backward continues
            h = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._inflight.append((idx, h))

    def finish(self):
        """Wait for the collectives, average, and scatter the reduced values back into .grad.
        Parameters that received no gradient this iteration are reduced as zeros."""
        if not self.active:
            return
        launched = {i for i, _ in self._inflight}
        for idx, b in enumerate(self._buckets):
            if idx not in launched:
                for p in b['params']:
                    i, off = self._where[p]
                    if p.grad is None:
                        b['flat'][off:off + p.numel()].zero_()
                    elif b['pending'] > 0:
                        b['flat'][off:off + p.numel()].copy_(p.grad.reshape(-1))
                self._inflight.append((idx, dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        for idx, h in self._inflight:
            h.wait()
            b = self._buckets[idx]
            b['flat'].div_(self.world)
            off = 0
            for p in b['params']:
                if p.grad is not None:
                    p.grad.copy_(b['flat'][off:off + p.numel()].view_as(p.grad))
                off += p.numel()
            b['pending'] = b['n']
        self._inflight = []

    def finish_flat(self):
        """Like finish(), but leaves the SUMMED gradients in the buckets and returns ({parameter: bucket slice}, 1 / world)
        for an optimizer that consumes them in place (afcm_amd.optim.FusedScrubAdam): no averaging pass and no copy back
        into .grad -- the scale rides in the optimizer kernel."""
        if not self.active:
            return None, 1.0
        launched = {i for i, _ in self._inflight}
        for idx, b in enumerate(self._buckets):
            if idx not in launched:
                for p in b['params']:
                    i, off = self._where[p]
                    if p.grad is None:
                        b['flat'][off:off + p.numel()].zero_()
                    elif b['pending'] > 0:
                        b['flat'][off:off + p.numel()].copy_(p.grad.reshape(-1))
                self._inflight.append((idx, dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        views = {}
        for idx, h in self._inflight:
            h.wait()
            b = self._buckets[idx]
            off = 0
            for p in b['params']:
                views[p] = b['flat'][off:off + p.numel()].view(p.shape)
                off += p.numel()
            b['pending'] = b['n']
        self._inflight = []
        return views, 1.0 / self.world

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
