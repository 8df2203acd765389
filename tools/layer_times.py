"""Per-geometry kernel times of the default bench configuration (instrumented pass, geometry in the labels)."""
import os, sys, types
os.environ['SG_PROFILE_DETAIL'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ga_gan_amd import _lib

args = types.SimpleNamespace(config=os.environ.get('CONFIG', 'c2'))
torch.cuda.set_device(0)
trainer, desc, res, batch_gpu = bench.make_trainer(args, 0, 1)
gen = torch.Generator(device=trainer.device).manual_seed(1)
real, real_c = trainer.synthetic_batch(gen)
for _ in range(3):
    trainer.step(real, real_c)
steps = 4
_lib.profile_begin()
for _ in range(steps):
    trainer.step(real, real_c)
rec = _lib.profile_end()
agg = {}
for name, nbytes, ms, flops in rec:
    a = agg.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1; a[1] += ms; a[2] += flops
tot = 0
for name, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f'{ms / steps:8.3f} ms/step {cnt / steps:6.1f}x {ms / cnt:7.3f} ms  {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s  {name}')
    tot += ms / steps
print('total listed', tot)
