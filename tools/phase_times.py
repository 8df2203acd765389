"""Diagnostic: time each training phase of a bench configuration on the GPU (prints as it goes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = bench.parse_args()
trainer, desc, res, batch_gpu = bench.make_trainer(args, 0, 1)
from ga_gan_amd.training import training_loop
real, real_c = trainer.synthetic_batch()
real = real.float() / 127.5 - 1
z = torch.randn(batch_gpu, 512, device=trainer.device)
print(desc, flush=True)
for rep in range(3):
    for phase in trainer.phases:
        phase.opt.zero_grad(set_to_none=True)
        training_loop.set_requires_grad(phase.module, trainer.requires_grad_parts[phase.name])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        trainer.loss.accumulate_gradients(phase=phase.name, real_img=real, real_c=real_c, gen_z=z, gen_c=real_c, sync=True, gain=phase.interval)
        torch.cuda.synchronize()
        print(f'rep {rep} {phase.name}: {(time.perf_counter() - t0) * 1e3:.1f} ms', flush=True)
        phase.module.requires_grad_(False)
print('peak mem GB', torch.cuda.max_memory_allocated() / 2**30, flush=True)
