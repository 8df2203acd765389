"""Times sg_epilogue_backward on layer shapes of the 1024 and 256 models (median of event-timed launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import layer_plugin as lp
cases = [(16, 64, 512, True), (16, 64, 512, False), (16, 32, 1024, True), (32, 128, 128, True), (32, 128, 128, False), (32, 512, 32, True), (64, 512, 32, False), (32, 256, 64, True)]
for n, c, h, gen in cases:
    for want_dz in (True, False):
        y = torch.randn([n, c, h, h], device='cuda').to(torch.bfloat16)
        dy = torch.randn_like(y)
        kw = dict(bias=torch.randn([c], device='cuda').to(torch.bfloat16), act='lrelu', gain=2 ** 0.5, clamp=256.0)
        if gen:
            kw.update(scale=torch.rand([n, c], device='cuda') + 0.5, noise=torch.randn([h, h], device='cuda'),
                      noise_gain=torch.full([], 0.1, device='cuda'))
        run = lambda: lp.epilogue_backward(dy, y, True, gen, gen, want_dz=want_dz, **kw)
        for _ in range(5): run()
        times = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): run()
            b.record(); torch.cuda.synchronize()
            times.append(a.elapsed_time(b) / 10)
        t = sorted(times)[2]
        byts = y.numel() * 2 * (3 if want_dz else 2)
        print(f'n{n} c{c} {h}^2 {"G" if gen else "D"} dz={int(want_dz)}  {t * 1e3:8.1f} us  {byts / t / 1e6:6.0f} GB/s', flush=True)
