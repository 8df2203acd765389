"""Reader for the fixtures written by tools/gen_golden.py: each .npz holds a JSON
list `meta` (one entry per case) and arrays named 'c<i>_<name>'."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Case:
    def __init__(self, meta, arrays):
        self.meta = meta
        self.arrays = arrays

    def __contains__(self, k):
        return k in self.arrays

    def np(self, k):
        return self.arrays[k]

    def t(self, k, device='cpu', dtype=None, grad=False):
        v = torch.from_numpy(np.array(self.arrays[k])).to(device)
        if dtype is not None and v.is_floating_point():
            v = v.to(dtype)
        return v.requires_grad_(grad)

    def get(self, k, **kw):
        return self.t(k, **kw) if k in self.arrays else None

    def group(self, prefix, device='cpu', dtype=None):
        """All arrays named '<prefix><rest>' as {rest: tensor}."""
        return {k[len(prefix):]: self.t(k, device=device, dtype=dtype) for k in self.arrays if k.startswith(prefix)}


def load(name):
    with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
        meta = json.loads(bytes(z['meta']).decode())
        per_case = [dict() for _ in meta]
        for key in z.files:
            if key == 'meta':
                continue
            idx, _, rest = key.partition('_')
            per_case[int(idx[1:])][rest] = z[key]
    return [Case(m, a) for m, a in zip(meta, per_case)]


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())
