"""Micro-benchmark: sg_conv2d_wgrad on the layer shapes of the 256 config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp

N = int(os.environ.get('N', 32))
LAYERS = [(64, 64, 256, torch.bfloat16), (128, 128, 128, torch.bfloat16), (256, 256, 64, torch.bfloat16),
          (512, 512, 32, torch.bfloat16), (512, 512, 16, torch.float32), (512, 512, 8, torch.float32)]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


LAYERS += [(512, 512, 64, torch.float32), (512, 512, 32, torch.float32)]
LAYERS += [(32, 32, 1024, torch.bfloat16), (64, 64, 512, torch.bfloat16)]   # 8, 9
LAYERS_STRIDED = [(64, 128, -264, torch.bfloat16), (128, 256, -136, torch.bfloat16), (256, 512, -72, torch.bfloat16)]   # h < 0: stride 2, input h x h
LAYERS += LAYERS_STRIDED
LAYERS += [(512, 512, -40, torch.bfloat16), (512, 512, 16, torch.bfloat16)]   # 13, 14: gradient planes of 16 x 16
if os.environ.get('ONLY'):
    LAYERS = [LAYERS[int(i)] for i in os.environ['ONLY'].split(',')]
for ci, co, h, dtype in LAYERS:
    stride = 1
    if h < 0:
        stride, h = 2, -h
        x = torch.randn([N, ci, h - 7, h], device='cuda').to(dtype)
        h = (h - 8) // 2
        g = torch.randn([N, co, h, h], device='cuda').to(dtype)
    else:
        x = torch.randn([N, ci, h, h], device='cuda').to(dtype)
        g = torch.randn([N, co, h, h], device='cuda').to(dtype)
    fn = lambda: cp.conv2d_wgrad(x, g, 3, 3, stride=stride, padding=1 if stride == 1 else 0)
    t = timeit(fn)
    flops = 2.0 * N * co * ci * 9 * h * h
    nbytes = (x.numel() + g.numel()) * x.element_size()
    print(f'wgrad {ci:4d}->{co:4d} @{h:4d} {str(dtype)[6:]:9s} {t:8.3f} ms ({flops / t / 1e9:7.1f} TF/s, {nbytes / t / 1e6:7.1f} GB/s)', flush=True)
