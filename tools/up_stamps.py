"""Diagnostic: per-phase cycle stamps of the one-pass up-sampling kernel (needs a library built with -DBAND_ABL_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import layer_plugin as lp, _lib
from ga_gan_amd.torch_utils.ops import upfirdn2d
N, ci, co, res = 32, 64, 32, 512
DT = torch.bfloat16
x = torch.randn(N, ci, res, res, device='cuda').to(DT)
w = (torch.randn(co, ci, 3, 3, device='cuda') / 24).to(DT)
s = 1 + 0.1 * torch.randn(N, ci, device='cuda'); d = 1 + 0.1 * torch.rand(N, co, device='cuda')
noise = torch.randn(2 * res, 2 * res, device='cuda'); ng = torch.tensor(0.1, device='cuda'); b = torch.zeros(co, device='cuda', dtype=DT)
f = upfirdn2d.setup_filter([1, 3, 3, 1], device='cuda')
m, fy4 = lp.separable_taps(f)
wf = lp.compose_up2_weights(w, m)
for _ in range(3):
    lp.conv_up2_fir(x, wf, co, fy4, in_scale=s, scale=d, noise=noise, noise_gain=ng, bias=b, act='lrelu', alpha=0.2, gain=1.414, clamp=256.0)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 1024)()
h = ctypes.CDLL(_lib.LIB_PATH)
assert h.sg_debug_up_stamps(buf) == 0
names = ['issue', 'late-fin', 'mult', 'fin', 'commit', 'barrier']
for wave in (0, 1, 4, 5):
    print('wave', wave)
    for st in range(1, 9):
        v = [buf[(wave * 16 + st) * 8 + k] for k in range(7)]
        prev = buf[(wave * 16 + st - 1) * 8 + 6]
        print('  step', st + 4, 'gap', v[0] - prev, ' '.join(f'{names[k]} {v[k + 1] - v[k]:5d}' for k in range(6)), 'total', v[6] - v[0])
