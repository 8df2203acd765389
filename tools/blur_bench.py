"""Micro-benchmark of the blur (upfirdn2d up=down=1) kernel on training shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import upfirdn2d_plugin as up
from ga_gan_amd.torch_utils.ops import upfirdn2d as U

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

f = U.setup_filter([1, 3, 3, 1], device='cuda')
for (n, c, h, w, pads, dt) in ((32, 32, 1025, 1032, (1, -6, 1, 1), torch.bfloat16), (16, 32, 1024, 1024, (2, 9, 2, 2), torch.bfloat16), (32, 64, 257, 264, (1, -6, 1, 1), torch.bfloat16), (32, 128, 129, 136, (1, -6, 1, 1), torch.bfloat16),
                               (32, 64, 256, 256, (2, 9, 2, 2), torch.bfloat16), (32, 256, 65, 72, (1, -6, 1, 1), torch.bfloat16),
                               (32, 512, 17, 24, (1, -6, 1, 1), torch.float32)):
    x = torch.randn(n, c, h, w, device='cuda').to(dt)
    fn = lambda: up.upfirdn2d(x, f, 1, 1, 1, 1, pads[0], pads[1], pads[2], pads[3], False, 4.0)
    y = fn(); t = timeit(fn)
    nbytes = (x.numel() + y.numel()) * x.element_size()
    print(f'blur {tuple(x.shape)} -> {tuple(y.shape)} {str(dt)[6:]:9s} {t:7.3f} ms  {nbytes / t / 1e6:7.1f} GB/s', flush=True)
for (n, c, h, w, dt) in ((16, 32, 1024, 1024, torch.bfloat16), (32, 64, 512, 512, torch.bfloat16), (64, 64, 256, 256, torch.bfloat16)):
    x = torch.randn(n, c, h, w, device='cuda').to(dt)
    fn = lambda: up.upfirdn2d(x, f, 1, 1, 2, 2, 1, 1, 1, 1, False, 1.0)
    y = fn(); t = timeit(fn)
    nbytes = (x.numel() + y.numel()) * x.element_size()
    print(f'down {tuple(x.shape)} -> {tuple(y.shape)} {str(dt)[6:]:9s} {t:7.3f} ms  {nbytes / t / 1e6:7.1f} GB/s', flush=True)
    fn = lambda: up.upfirdn2d(y, f, 2, 2, 1, 1, 2, 1, 2, 1, False, 4.0)
    z = fn(); t = timeit(fn)
    nbytes = (z.numel() + y.numel()) * x.element_size()
    print(f'up   {tuple(y.shape)} -> {tuple(z.shape)} {str(dt)[6:]:9s} {t:7.3f} ms  {nbytes / t / 1e6:7.1f} GB/s', flush=True)
