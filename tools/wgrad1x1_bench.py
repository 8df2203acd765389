"""Times the weight gradient of the discriminator's 1x1 skip layers (bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp
for n, ci, co, h in [(64, 64, 128, 128), (64, 128, 256, 64), (64, 256, 512, 32), (16, 32, 64, 512), (16, 64, 128, 256)]:
    x = torch.randn([n, ci, h, h], device='cuda').to(torch.bfloat16)
    dy = torch.randn([n, co, h, h], device='cuda').to(torch.bfloat16)
    run = lambda: cp.conv2d_wgrad(x, dy, 1, 1)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10
    gb = n * (ci + co) * h * h * 2 / 1e9
    print(f'n{n} {ci}->{co} @{h}: {t:.3f} ms  {gb / t * 1e3:6.0f} GB/s  {2.0 * n * ci * co * h * h / t / 1e9:5.0f} TF/s', flush=True)
