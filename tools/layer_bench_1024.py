"""Kernel-level headline: the modulated-convolution + upfirdn2d layers of the 1024^2 generator at resolutions of
256^2 and above, batch 32, bfloat16, forward and backward, priced against the HBM roofline.

Algorithmic bytes (ideal fusion, per image): forward (Cin*Hin^2 + Cout*Hout^2)*2; data gradient the same with the
saved output read once more for the activation mask; weight gradient reads x and dy once.  Peak 8 TB/s.
    python tools/layer_bench_1024.py [N]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd.torch_utils.ops import fused_layer, upfirdn2d

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DEV, DT, PEAK = 'cuda', torch.bfloat16, 8000.0
LAYERS = [  # name, cin, cout, input resolution, up, kernel
    ('b256.conv0', 256, 128, 128, 2, 3), ('b256.conv1', 128, 128, 256, 1, 3),
    ('b512.conv0', 128, 64, 256, 2, 3), ('b512.conv1', 64, 64, 512, 1, 3),
    ('b1024.conv0', 64, 32, 512, 2, 3), ('b1024.conv1', 32, 32, 1024, 1, 3),
    ('b1024.torgb', 32, 3, 1024, 1, 1),
]


def timeit(fn, iters=5):
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


f = upfirdn2d.setup_filter([1, 3, 3, 1], device=DEV)
print(f'{"layer":12s} {"fwd ms":>8s} {"GB/s":>7s} {"%HBM":>6s} {"TF/s":>6s} | {"bwd ms":>8s} {"GB/s":>7s} {"%HBM":>6s} {"TF/s":>6s}')
tot = [0.0, 0.0, 0.0, 0.0]
for name, ci, co, res, up, k in LAYERS:
    out = res * up
    x = torch.randn(N, ci, res, res, device=DEV).to(DT).requires_grad_()
    w = (torch.randn(co, ci, k, k, device=DEV) / (ci * k * k) ** 0.5).to(DT).requires_grad_()
    styles = (1 + 0.1 * torch.randn(N, ci, device=DEV)).requires_grad_()
    dcoefs = (1 + 0.1 * torch.rand(N, co, device=DEV)).requires_grad_() if k == 3 else None
    noise = torch.randn(out, out, device=DEV) if k == 3 else None
    strength = torch.tensor(0.1, device=DEV, requires_grad=True) if k == 3 else None
    bias = torch.zeros(co, device=DEV, dtype=DT, requires_grad=True)
    act = 'lrelu' if k == 3 else 'linear'

    def fwd():
        return fused_layer.conv_layer(x, w, f=f, up=up, padding=k // 2, flip_weight=(up == 1), styles=styles, dcoefs=dcoefs,
                                      noise=noise, noise_strength=strength, bias=bias, act=act, clamp=256.0)
    y = fwd()
    dy = torch.randn_like(y)

    def fwd_bwd():
        torch.autograd.grad(fwd(), [x, w, styles], dy)
    t_f = timeit(fwd)
    t_b = timeit(fwd_bwd) - t_f
    bytes_f = N * (ci * res * res + co * out * out) * 2
    bytes_b = bytes_f + N * co * out * out * 2 + N * (ci * res * res + co * out * out) * 2   # dgrad (+ saved y) and wgrad
    flops_f = 2.0 * N * ci * co * k * k * res * res + (2.0 * 16 * N * co * out * out if up == 2 else 0)
    print(f'{name:12s} {t_f:8.3f} {bytes_f / t_f / 1e6:7.0f} {100 * bytes_f / t_f / 1e6 / PEAK:6.1f} {flops_f / t_f / 1e9:6.0f} | '
          f'{t_b:8.3f} {bytes_b / t_b / 1e6:7.0f} {100 * bytes_b / t_b / 1e6 / PEAK:6.1f} {2 * flops_f / t_b / 1e9:6.0f}', flush=True)
    tot[0] += t_f; tot[1] += bytes_f; tot[2] += t_b; tot[3] += bytes_b
    del x, w, y, dy
    torch.cuda.empty_cache()
print(f'{"all":12s} {tot[0]:8.3f} {tot[1] / tot[0] / 1e6:7.0f} {100 * tot[1] / tot[0] / 1e6 / PEAK:6.1f}        | {tot[2]:8.3f} {tot[3] / tot[2] / 1e6:7.0f} {100 * tot[3] / tot[2] / 1e6 / PEAK:6.1f}')
