"""Micro-benchmark: the 1x1 residual-skip convolutions of the 1024^2 discriminator (forward and data gradient), bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp

N = int(os.environ.get('N', 32))
SHAPES = [(32, 64, 512), (64, 128, 256), (128, 256, 128), (256, 512, 64), (64, 32, 512), (128, 64, 256), (256, 128, 128), (512, 256, 64)]


def timeit(fn, iters=10):
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


for ci, co, res in SHAPES:
    x = torch.randn(N, ci, res, res, device='cuda').to(torch.bfloat16)
    w = (torch.randn(co, ci, 1, 1, device='cuda') / ci ** 0.5).to(torch.bfloat16)
    plan = cp.plan_conv(1, 1)
    packed = plan.pack(w)
    t = timeit(lambda: cp.conv2d(x, packed, plan, co, gain=0.7071, clamp=256.0))
    gb = N * (ci + co) * res * res * 2 / 1e9
    print(f'{cp.kernel_name(x, plan, co):20s} {ci:4d}->{co:4d} @{res:4d} n{N}: {t * 1e3:7.1f} us  {gb / t:7.1f} TB/s... {gb / t * 1000:7.0f} GB/s  {2 * N * ci * co * res * res / t / 1e9:6.0f} TF/s', flush=True)
    del x

if os.environ.get('DOWN'):   # the strided 3x3 layer behind the blur of the first discriminator block
    import torch.nn.functional as F
    for n in (16, 32):
        x = torch.randn(n, 32, 1025, 1032, device='cuda').to(torch.bfloat16)
        w = (torch.randn(64, 32, 3, 3, device='cuda') / 17).to(torch.bfloat16)
        plan = cp.plan_conv(3, 3, stride=2, padding=0)
        packed = plan.pack(w)
        t = timeit(lambda: cp.conv2d(x, packed, plan, 64, out_hw=(512, 512), act='lrelu', gain=0.7071, clamp=256.0))
        gb = (x.numel() + n * 64 * 512 * 512) * 2 / 1e9
        print(f'{cp.kernel_name(x, plan, 64, out_hw=(512, 512)):24s} 32->64 s2 out512 n{n}: {t * 1e3:7.1f} us {gb / t * 1000:7.0f} GB/s {2 * n * 32 * 64 * 9 * 512 * 512 / t / 1e9:6.0f} TF/s', flush=True)
        del x
