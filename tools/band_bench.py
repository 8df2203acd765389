"""Timing of the same-size 3x3 layers at 1024^2 / 512^2 (batch N) through sg_conv2d: forward with the full layer
epilogue, and the data-gradient call (masked input, style-gradient reduction).  python tools/band_bench.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DEV, DT = 'cuda', torch.bfloat16


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


SHAPES = ((32, 32, 1024), (32, 64, 512), (64, 64, 512))
WID = int(os.environ.get('BAND_W', 0))
if os.environ.get('BAND_ONLY'):
    SHAPES = SHAPES[:1]
for ci, co, res in SHAPES:
    wid = WID or res
    x = torch.randn(N, ci, res, wid, device=DEV).to(DT)
    w = (torch.randn(co, ci, 3, 3, device=DEV) / (ci * 9) ** 0.5).to(DT)
    s = 1 + 0.1 * torch.randn(N, ci, device=DEV)
    d = 1 + 0.1 * torch.rand(N, co, device=DEV)
    noise = torch.randn(res, wid, device=DEV)
    ng = torch.tensor(0.1, device=DEV)
    b = torch.zeros(co, device=DEV, dtype=DT)
    plan = cp.plan_conv(3, 3, padding=1)
    packed = plan.pack(w)
    y = torch.empty(N, co, res, wid, device=DEV, dtype=DT)
    fwd = lambda: cp.conv2d(x, packed, plan, co, in_scale=s, out_scale=d, bias=b, noise=noise, noise_gain=ng, act='lrelu',
                            gain=1.414, clamp=256.0, out=y)
    t = timeit(fwd)
    nbytes = N * (ci + co) * res * wid * 2
    flops = 2.0 * N * ci * co * 9 * res * wid
    print(f'{ci}->{co} @{res} fwd   {t:7.3f} ms  {nbytes / t / 1e6:6.0f} GB/s ({nbytes / t / 8e7:4.1f}% HBM)  {flops / t / 1e9:5.0f} TF/s', flush=True)
    if ci == co:
        dy = torch.randn_like(y)
        planT = cp.plan_conv_transpose(3, 3, stride=1, padding=1)
        packedT = planT.pack(w, transposed_layout=True)
        dx = torch.empty_like(x)
        rs = torch.empty(N, ci, device=DEV)
        if cp.supports_mask(dy, planT, ci, reduce=True):
            bwd = lambda: cp.conv2d(dy, packedT, planT, ci, in_scale=d, out_scale=s, gain=1.414, reduce_x=x, mask_y=y,
                                    mask_alpha=0.2, mask_clamp=256.0, out=dx, reduce_sum=rs)
            t = timeit(bwd)
            nbytes = N * (3 * co + 2 * ci) * res * res * 2   # dy, y, x read; dx written ... x counted once
            nbytes = N * (2 * co + 2 * ci) * res * wid * 2
            print(f'{ci}->{co} @{res} dgrad {t:7.3f} ms  {nbytes / t / 1e6:6.0f} GB/s ({nbytes / t / 8e7:4.1f}% HBM)  {flops / t / 1e9:5.0f} TF/s', flush=True)
    del x, y
