"""Micro-benchmark: per-plane scale and dot product at the sizes of the 1024^2 generator's regularisation pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd.torch_utils.ops import planes


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for n, c, r in ((8, 32, 1024), (8, 64, 512), (8, 128, 256), (16, 64, 256), (16, 512, 32)):
    x = torch.randn(n, c, r, r, device='cuda').to(torch.bfloat16)
    y = torch.randn_like(x)
    s = torch.randn(n, c, device='cuda')
    with torch.no_grad():
        t1 = timeit(lambda: planes.scale_planes(x, s))
        t2 = timeit(lambda: planes.dot_planes(x, y))
        t3 = timeit(lambda: x * s.to(x.dtype).reshape(n, c, 1, 1))
        t4 = timeit(lambda: (x * y).sum([2, 3]))
    gb = x.numel() * 2 / 1e9
    print(f'[{n},{c},{r},{r}] scale {t1 * 1e3:7.1f} us {2 * gb / t1:6.2f} TB/s (torch {t3 * 1e3:7.1f} us)   dot {t2 * 1e3:7.1f} us {2 * gb / t2:6.2f} TB/s (torch {t4 * 1e3:7.1f} us)', flush=True)
