"""Which source lines issue the torch operators of a training step (forward side and Python-level backward code on
the calling thread): runs the default bench configuration under a dispatch mode that records, per (operator,
innermost line of this package), calls per step and the largest output it produced."""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

table = collections.defaultdict(lambda: [0, 0])


class Recorder(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        where = '?'
        for frame in reversed(traceback.extract_stack(limit=40)):
            if ('ga-gan_amd' in frame.filename or 'ga_gan_amd' in frame.filename):
                where = f'{frame.filename.split("gan_amd/")[-1]}:{frame.lineno} {frame.line[:70]}'
                break
        numel = out.numel() if isinstance(out, torch.Tensor) else 0
        entry = table[(str(func).replace('aten.', ''), where)]
        entry[0] += 1
        entry[1] = max(entry[1], numel)
        return out


args = argparse.Namespace(config='c2', ada=False, graphs=False)
torch.cuda.set_device(0)
trainer, desc, res, batch = bench.make_trainer(args, 0, 1)
real = torch.randint(0, 256, [batch, 3, res, res], dtype=torch.uint8, device='cuda')
real_c = torch.zeros([batch, 0], device='cuda')
for _ in range(16):
    trainer.step(real, real_c)
torch.cuda.synchronize()
STEPS = 16
with Recorder():
    for _ in range(STEPS):
        trainer.step(real, real_c)
torch.cuda.synchronize()
skip = ('view', 'reshape', 'detach', 'alias', 'expand', 'unsqueeze', 'squeeze', 'slice', 'select', 't.default', 'transpose', 'permute',
        'empty', 'as_strided', 'unbind', 'split', '_unsafe_view', 'unflatten', 'is_', 'lift_fresh', 'set_', 'record_stream')
rows = [(k, v) for k, v in table.items() if not any(k[0].startswith(s) for s in skip)]
rows.sort(key=lambda kv: -kv[1][0])
print(f'{sum(v[0] for _, v in rows) / STEPS:.0f} non-view operator calls per step on the calling thread')
for (name, where), (n, numel) in rows[:90]:
    print(f'{n / STEPS:7.1f} max {numel:>10d}  {name[:28]:28s} {where[:125]}')
