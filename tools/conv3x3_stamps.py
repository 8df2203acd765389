"""Diagnostic: cycles per phase of conv3x3_kernel's K loop (needs the stamped build tools/_variants/lib_stamps.so, made by
tools/build_variants.sh - the directory is listed in .gpurunignore, take it out of there for such a run;
SG_OPS_LIBRARY must point at the library).  GEO=n,ci,co,h"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp, _lib
n, ci, co, h = [int(v) for v in os.environ.get('GEO', '32,512,512,32').split(',')]
x = torch.randn([n, ci, h, h], device='cuda').to(torch.bfloat16)
w = (torch.randn([co, ci, 3, 3], device='cuda') / 20).to(torch.bfloat16)
s = (torch.rand([n, ci], device='cuda') + 0.5) if os.environ.get('MOD') else None
plan = cp.plan_conv(3, 3, padding=1); packed = plan.pack(w)
for _ in range(200):          # long enough for the clock to settle
    y = cp.conv2d(x, packed, plan, co, in_scale=s)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 2 * 8))()
lib = ctypes.CDLL(_lib.LIB_PATH)
assert lib.sg_debug_conv3x3_stamps(buf) == 0
if os.environ.get('RESIDENT'):      # the weights-resident kernel (tools/ablation/resident_stamps_build.py): waves 0 and 2
    names = ['barrier 1', 'commit', 'barrier 2', 'plan + operands', 'multiply + loads', 'epilogue']
    for half in (0, 1):
        rows = [[buf[(wg * 2 + half) * 8 + k] for k in range(8)] for wg in range(64)]
        rows = [r for r in rows if r[7] > 0]
        med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]
        tiles = med(7)
        print(f'wave {2 * half}: {tiles} tiles, whole kernel {med(6)} cycles = {med(6) / tiles:.0f} per tile;  ' +
              '  '.join(f'{names[k]} {med(k) / tiles:6.0f}' for k in range(6)))
    sys.exit(0)
names = ['barrier 1', 'commit', 'barrier 2', 'fetch', 'multiply']
for half in (0, 1):
    rows = [[buf[(wg * 2 + half) * 8 + k] for k in range(8)] for wg in range(64)]
    rows = [r for r in rows if r[7] > 0]
    chunks = rows[0][7]
    med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]
    clock = med(5) / max(med(6), 1) * 100e6 / 1e9
    print(f'wave {4 * half}: {chunks} chunks in the loop, whole kernel {med(5)} cycles, clock {clock:.2f} GHz;  per chunk: ' +
          '  '.join(f'{names[k]} {med(k) / chunks:6.0f}' for k in range(5)) + f'   sum {sum(med(k) for k in range(5)) / chunks:6.0f}')
