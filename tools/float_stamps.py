"""Diagnostic: cycles per phase of the stage loop of conv2d_mfma_kernel's float path (needs the stamped build of
tools/ablation/float_stamps_build.py; SG_OPS_LIBRARY must point at it).  GEO=n,ci,co,h"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp, _lib
n, ci, co, h = [int(v) for v in os.environ.get('GEO', '16,512,512,64').split(',')]
x = torch.randn([n, ci, h, h], device='cuda')
w = torch.randn([co, ci, 3, 3], device='cuda') / 20
plan = cp.plan_conv(3, 3, padding=1); packed = plan.pack(w)
for _ in range(100):
    y = cp.conv2d(x, packed, plan, co)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 2 * 8))()
lib = ctypes.CDLL(_lib.LIB_PATH)
assert lib.sg_debug_conv2d_stamps(buf) == 0
names = ['issue', 'multiply', 'commit', 'copies landed', 'barrier']
for half in (0, 1):
    rows = [[buf[(wg * 2 + half) * 8 + k] for k in range(8)] for wg in range(64)]
    rows = [r for r in rows if r[6] > 0]
    med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]
    stages = med(6)
    print(f'wave {4 * half}: {stages} stages, from the first issue to the end of the loop {med(5)} cycles = {med(5) / stages:.0f} per stage;  ' +
          '  '.join(f'{names[k]} {med(k) / stages:6.0f}' for k in range(5)))
