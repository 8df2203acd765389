"""Diagnostic: which framework operators (not this library's kernels) take device time in the training phases of a
bench configuration, by operator, input shapes and the Python frame that issued them.

    python tools/op_profile.py --config c3 > gpurun_out/op_profile_c3.txt
"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

args = bench.parse_args()
trainer, desc, res, batch_gpu = bench.make_trainer(args, 0, 1)
from ga_gan_amd.training import training_loop
real, real_c = trainer.synthetic_batch()
real = real.float() / 127.5 - 1
z = torch.randn(batch_gpu, 512, device=trainer.device)
print(desc, flush=True)


def run_phase(phase):
    phase.opt.zero_grad(set_to_none=True)
    training_loop.set_requires_grad(phase.module, trainer.requires_grad_parts[phase.name])
    trainer.loss.accumulate_gradients(phase=phase.name, real_img=real, real_c=real_c, gen_z=z, gen_c=real_c, sync=True, gain=phase.interval)
    phase.module.requires_grad_(False)


for phase in trainer.phases:      # warm-up
    run_phase(phase)
torch.cuda.synchronize()

for phase in trainer.phases:
    if phase.name not in os.environ.get('PHASES', 'Gmain,Dmain').split(','):
        continue
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        run_phase(phase)
        torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        dev = getattr(ev, 'self_device_time_total', 0) or 0
        if dev <= 0 or not ev.name.startswith('aten::'):
            continue
        frame = next((s for s in (ev.stack or []) if 'ga-gan_amd' in s or 'ga_gan_amd' in s or 'bench' in s), '(autograd)')
        frame = frame.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '')
        shapes = str([s for s in (ev.input_shapes or []) if s])[:90]
        key = (ev.name, shapes, frame[:110])
        rows[key][0] += dev
        rows[key][1] += 1
    total = sum(v[0] for v in rows.values())
    kernels = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            name = ev.name.replace('(anonymous namespace)::', '').split('(')[0][:100]
            kernels[name][0] += ev.device_time_total if hasattr(ev, 'device_time_total') else 0
            kernels[name][1] += 1
    print(f'\n== {phase.name}: kernels {sum(v[0] for v in kernels.values()) / 1e3:.2f} ms of device time', flush=True)
    for name, (us, n) in sorted(kernels.items(), key=lambda kv: -kv[1][0])[:32]:
        print(f'{us / 1e3:8.3f} ms {n:4d}x {name}')
    print(f'\n== {phase.name}: framework operators {total / 1e3:.2f} ms of device time', flush=True)
    for (name, shapes, frame), (us, n) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f'{us / 1e3:8.3f} ms {n:4d}x {name:28s} {shapes:90s} {frame}')
