"""Micro-benchmark: sg_conv2d vs the library convolution on the layer shapes of a config."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ga_gan_amd import conv2d_plugin as cp

DEV = 'cuda'
N = int(os.environ.get('N', 32))
LAYERS = [  # ci, co, h, dtype, kind
    (64, 64, 256, torch.bfloat16, 'conv'), (128, 128, 128, torch.bfloat16, 'conv'),
    (256, 256, 64, torch.bfloat16, 'conv'), (512, 512, 32, torch.bfloat16, 'conv'),
    (512, 512, 16, torch.float32, 'conv'), (512, 512, 8, torch.float32, 'conv'),
    (128, 64, 128, torch.bfloat16, 'up'), (256, 128, 64, torch.bfloat16, 'up'),
    (64, 3, 256, torch.bfloat16, '1x1'), (3, 64, 256, torch.bfloat16, '1x1'),
    (64, 128, 257, torch.bfloat16, 'down'),
    (128, 64, 128, torch.bfloat16, 'up_t'), (256, 128, 64, torch.bfloat16, 'up_t'), (512, 256, 32, torch.bfloat16, 'up_t'),
    (64, 128, 264, torch.bfloat16, 'down_t'), (128, 256, 136, torch.bfloat16, 'down_t'), (256, 512, 72, torch.bfloat16, 'down_t'),
]
if os.environ.get('LOWRES'):   # the float32 layers of a 256 x 256 network
    f = torch.float32
    LAYERS = [(512, 512, 16, f, 'conv'), (512, 512, 8, f, 'conv'), (512, 512, 4, f, 'conv'), (512, 512, 8, f, 'up_t'),
              (512, 512, 4, f, 'up_t'), (512, 512, 24, f, 'down_t'), (512, 512, 16, f, 'down_t'), (512, 512, 16, f, 'up'),
              (512, 512, 8, f, 'up'), (512, 512, 4, f, 'up')]
if os.environ.get('F32BIG'):   # the float32 layers of a 1024 x 1024 network (N=16)
    LAYERS = [(512, 512, 64, torch.float32, 'conv'), (512, 512, 32, torch.float32, 'conv'), (512, 512, 32, torch.float32, 'up_t'), (512, 512, 72, torch.float32, 'down_t')]
if os.environ.get('ONLY'):
    LAYERS = [LAYERS[int(i)] for i in os.environ['ONLY'].split(',')]
if os.environ.get('BIG'):
    LAYERS = [(32, 32, 1024, torch.bfloat16, 'conv'), (64, 64, 512, torch.bfloat16, 'conv'), (64, 32, 512, torch.bfloat16, 'up')]


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


for ci, co, h, dtype, kind in LAYERS:
    out_hw = None
    if kind == 'up_t':     # as in training: output rows padded to whole packets
        kind, out_hw = 'up', (2 * h + 1, (2 * h + 1 + 7) // 8 * 8)
        x = torch.randn([N, ci, h, h], device=DEV).to(dtype)
    elif kind == 'down_t':   # as in training: blurred plane of (h - 7) rows, h columns (rows padded to whole packets)
        kind, out_hw = 'down', ((h - 8) // 2, (h - 8) // 2)
        x = torch.randn([N, ci, h - 7, h], device=DEV).to(dtype)
    else:
        x = torch.randn([N, ci, h, h], device=DEV).to(dtype)
    if kind == 'conv':
        w = torch.randn([co, ci, 3, 3], device=DEV).to(dtype) / 50
        plan = cp.plan_conv(3, 3, padding=1); packed = plan.pack(w)
        lib = lambda: F.conv2d(x, w, padding=1)
        taps = 9
    elif kind == '1x1':
        w = torch.randn([co, ci, 1, 1], device=DEV).to(dtype) / 50
        plan = cp.plan_conv(1, 1); packed = plan.pack(w)
        lib = lambda: F.conv2d(x, w)
        taps = 1
    elif kind == 'down':
        w = torch.randn([co, ci, 3, 3], device=DEV).to(dtype) / 50
        plan = cp.plan_conv(3, 3, stride=2); packed = plan.pack(w)
        lib = lambda: F.conv2d(x, w, stride=2)
        taps = 9
    else:
        w = torch.randn([ci, co, 3, 3], device=DEV).to(dtype) / 50
        plan = cp.plan_conv_transpose(3, 3, stride=2); packed = plan.pack(w, transposed_layout=True)
        lib = lambda: F.conv_transpose2d(x, w, stride=2)
        taps = 9 / 4
    ours = lambda: cp.conv2d(x, packed, plan, co, out_hw=out_hw)
    y = ours()
    if os.environ.get('NOLIB'):
        lib = ours
    t0 = time.time(); lib(); torch.cuda.synchronize(); first = time.time() - t0
    t_ours, t_lib = timeit(ours), timeit(lib)
    oh = y.shape[2]
    flops = 2.0 * N * co * ci * taps * oh * oh
    nbytes = (x.numel() + y.numel()) * x.element_size()
    print(f'{kind:5s} {ci:4d}->{co:4d} @{h:4d} {str(dtype)[6:]:9s} ours {t_ours:8.3f} ms ({flops / t_ours / 1e9:7.1f} TF/s, {nbytes / t_ours / 1e6:7.1f} GB/s)'
          f'   lib {t_lib:8.3f} ms ({flops / t_lib / 1e9:7.1f} TF/s)  lib first call {first:6.2f}s', flush=True)
