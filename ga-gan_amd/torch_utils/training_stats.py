"""Minimal running statistics: `report(name, value)` accumulates [count, sum, sum of
squares] per name on the value's device; `Collector` reads the means, summed over
ranks with one all-reduce per update (reference: torch_utils/training_stats.py)."""
import re

import torch

_counters = {}   # name -> float64 tensor [3] on the reporting device
_powers = {}     # device -> float64 [[0], [1], [2]]
_enabled = True


def enable(flag):
    global _enabled
    _enabled = bool(flag)


def report(name, value):
    if not _enabled:
        return value
    elems = torch.as_tensor(value).detach().flatten().to(torch.float64)
    if elems.numel() == 0:
        return value
    powers = _powers.get(elems.device)
    if powers is None:
        powers = torch.arange(3, dtype=torch.float64, device=elems.device).reshape(3, 1)   # (a kernel, not a copy: legal under stream capture)
        if not (elems.is_cuda and torch.cuda.is_current_stream_capturing()):
            _powers[elems.device] = powers
    moments = elems.unsqueeze(0).pow(powers).sum(dim=1)    # count, sum, sum of squares: two kernels
    if name in _counters and _counters[name].device == moments.device:
        _counters[name] += moments
    else:
        _counters[name] = moments
    return value


class Collector:
    def __init__(self, regex='.*'):
        self._regex = re.compile(regex)
        self._means = {}

    def update(self):
        names = sorted(n for n in _counters if self._regex.fullmatch(n))
        if not names:
            return
        stacked = torch.stack([_counters[n] for n in names])
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(stacked)
        stacked = stacked.cpu()
        for n, row in zip(names, stacked):
            self._means[n] = float(row[1] / row[0]) if float(row[0]) > 0 else float('nan')
            _counters[n].zero_()

    def __getitem__(self, name):
        return self._means.get(name, float('nan'))

    def names(self):
        return sorted(self._means)
