"""Live per-kernel-family timing with HIP events (torch.cuda.Event on the stream the kernels are launched on).

`bench.py` switches this on for the timed region: each hot-op launch is bracketed by two events and tagged with
its algorithmic work (bytes for the HBM-bound FIR ops, flops for the MFMA convs), so the roofline fraction is
computed from launch durations measured in the run itself."""
import collections

import torch

enabled = False
_records = []   # (family, start_event, end_event, work)


def start():
    global enabled
    _records.clear()
    enabled = True


def stop():
    global enabled
    enabled = False


class _Span:
    __slots__ = ('family', 'work', 'ev0')

    def __init__(self, family, work):
        self.family, self.work = family, work
        self.ev0 = torch.cuda.Event(enable_timing=True)
        self.ev0.record()

    def end(self):
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        _records.append((self.family, self.ev0, ev1, self.work))


def span(family, work):
    return _Span(family, work) if enabled else None


def summary():
    """{family: dict(launches, total_ms, work)} -- call after torch.cuda.synchronize()."""
    out = collections.OrderedDict()
    for fam, e0, e1, work in _records:
        d = out.setdefault(fam, dict(launches=0, total_ms=0.0, work=0.0))
        d['launches'] += 1
        d['total_ms'] += e0.elapsed_time(e1)
        d['work'] += work
    return out
