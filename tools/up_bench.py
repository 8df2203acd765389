"""Timing of the one-pass up-sampling layers (sg_conv_up2_fir) of the 1024^2 generator at batch N, next to the
two-pass form (transposed convolution, then filter + epilogue).  python tools/up_bench.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp, layer_plugin as lp
from ga_gan_amd.torch_utils.ops import upfirdn2d

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


SHAPES = ((64, 32, 512), (128, 64, 256), (256, 128, 128))
if os.environ.get('UP_ONLY'):
    SHAPES = SHAPES[:1]
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=DEV)
m, fy4 = lp.separable_taps(f)
for ci, co, res in SHAPES:
    x = torch.randn(N, ci, res, res, device=DEV).to(DT)
    w = (torch.randn(co, ci, 3, 3, device=DEV) / (ci * 9) ** 0.5).to(DT)
    s = 1 + 0.1 * torch.randn(N, ci, device=DEV)
    d = 1 + 0.1 * torch.rand(N, co, device=DEV)
    noise = torch.randn(2 * res, 2 * res, device=DEV)
    ng = torch.tensor(0.1, device=DEV)
    b = torch.zeros(co, device=DEV, dtype=DT)
    ep = dict(scale=d, noise=noise, noise_gain=ng, bias=b, act='lrelu', alpha=0.2, gain=1.414, clamp=256.0)
    wf = lp.compose_up2_weights(w, m)
    t1 = timeit(lambda: lp.conv_up2_fir(x, wf, co, fy4, in_scale=s, **ep))
    nbytes = N * (ci * res * res + co * 4 * res * res) * 2
    flops = 2.0 * N * ci * co * 18 * res * res
    line = f'{ci}->{co} @{res}->{2 * res} one pass {t1:7.3f} ms  {nbytes / t1 / 1e6:6.0f} GB/s ({nbytes / t1 / 8e7:4.1f}% HBM)  {flops / t1 / 1e9:5.0f} TF/s'
    if not os.environ.get('UP_ONLY'):
        plan = cp.plan_conv_transpose(3, 3, stride=2, padding=0)
        packed = plan.pack(w)
        extra = (-(2 * res + 1)) % 8

        def two():
            t = cp.conv2d(x, packed, plan, co, in_scale=s, out_hw=(2 * res + 1, 2 * res + 1 + extra))
            return lp.blur_epilogue(t, f, (1, 1 - extra, 1, 1), False, 4.0, **ep)
        t2 = timeit(two)
        line += f' | two passes {t2:7.3f} ms'
    print(line, flush=True)
    del x
