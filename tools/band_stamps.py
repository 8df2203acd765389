"""Diagnostic: per-phase cycle stamps of the band kernel (needs a library built with -DBAND_ABL_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp, _lib
N, ci, co, res = 32, 32, 32, 1024
x = torch.randn(N, ci, res, res, device='cuda').to(torch.bfloat16)
w = (torch.randn(co, ci, 3, 3, device='cuda') / 17).to(torch.bfloat16)
s = 1 + 0.1 * torch.randn(N, ci, device='cuda'); d = 1 + 0.1 * torch.rand(N, co, device='cuda')
noise = torch.randn(res, res, device='cuda'); ng = torch.tensor(0.1, device='cuda'); b = torch.zeros(co, device='cuda', dtype=torch.bfloat16)
plan = cp.plan_conv(3, 3, padding=1); packed = plan.pack(w); y = torch.empty(N, co, res, res, device='cuda', dtype=torch.bfloat16)
for _ in range(3):
    cp.conv2d(x, packed, plan, co, in_scale=s, out_scale=d, bias=b, noise=noise, noise_gain=ng, act='lrelu', gain=1.414, clamp=256.0, out=y)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
h = ctypes.CDLL(_lib.LIB_PATH)
assert h.sg_debug_band_stamps(buf) == 0
names = ['issue', 'late-epi', 'mult', 'epi', 'commit', 'barrier']
for wave in (0, 1, 2, 4, 5, 6):
    print('wave', wave)
    for st in range(1, 7):
        v = [buf[(wave * 8 + st) * 8 + k] for k in range(7)]
        prev = buf[(wave * 8 + st - 1) * 8 + 6]
        print('  step', st + 4, 'gap', v[0] - prev, ' '.join(f'{names[k]} {v[k + 1] - v[k]:5d}' for k in range(6)), 'total', v[6] - v[0])
