"""Times a few 3x3 convolution geometries of the narrow layers (median of event-timed launches); run once per library
(SG_OPS_LIBRARY) to compare two builds on one box.  CASES="n,ci,co,h[,mod]; ..." overrides the list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp
cases = os.environ.get('CASES', '16,64,64,512; 16,64,64,512,mod; 32,64,64,256; 16,32,32,1024; 16,128,64,512; 16,64,128,256')
for case in cases.split(';'):
    f = case.strip().split(',')
    n, ci, co, h = [int(v) for v in f[:4]]
    mod = len(f) > 4
    x = torch.randn([n, ci, h, h], device='cuda').to(torch.bfloat16)
    w = (torch.randn([co, ci, 3, 3], device='cuda') / 20).to(torch.bfloat16)
    s = (torch.rand([n, ci], device='cuda') + 0.5) if mod else None
    plan = cp.plan_conv(3, 3, padding=1); packed = plan.pack(w)
    for _ in range(30): cp.conv2d(x, packed, plan, co, in_scale=s)
    times = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): cp.conv2d(x, packed, plan, co, in_scale=s)
        b.record(); torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / 20)
    flops = 2.0 * n * co * ci * 9 * h * h
    byts = 2.0 * n * (ci + co) * h * h
    t = sorted(times)[2]
    print(f'{case.strip():24s} {t:7.3f} ms  {flops / t / 1e9:7.0f} TFLOP/s  {byts / t / 1e6:6.0f} GB/s', flush=True)
