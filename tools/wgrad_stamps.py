"""Diagnostic: per-phase cycle sums of the float weight-gradient kernel (needs a library built with -DBAND_ABL_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp, _lib
N, c, res = 32, 512, 64
x = torch.randn(N, c, res, res, device='cuda')
g = torch.randn(N, c, res, res, device='cuda')
for _ in range(3):
    cp.conv2d_wgrad(x, g, 3, 3, stride=1, padding=1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
h = ctypes.CDLL(_lib.LIB_PATH)
assert h.sg_debug_wgrad_stamps(buf) == 0
for wave in range(4):
    v = [buf[wave * 16 + k] for k in range(6)]
    n = max(v[3], 1)
    print(f'wave {wave}: tiles {v[3]}  per tile: stage {v[0] / n:8.0f}  barrier {v[1] / n:8.0f}  compute {v[2] / n:8.0f}   epilogue {v[4]}  loop {v[5]}  (clocks of s_memtime, 100 MHz)')
