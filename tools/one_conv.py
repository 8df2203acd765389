"""One convolution geometry, a few launches: the subject of a counter-collection run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp
n, ci, co, h = [int(v) for v in os.environ.get('GEO', '16,32,32,1024').split(',')]
dtype = dict(bf16=torch.bfloat16, f16=torch.float16, f32=torch.float32)[os.environ.get('DTYPE', 'bf16')]
x = torch.randn([n, ci, h, h], device='cuda').to(dtype)
w = (torch.randn([co, ci, 3, 3], device='cuda') / 20).to(dtype)
plan = cp.plan_conv(3, 3, padding=1); packed = plan.pack(w)
for _ in range(3):
    y = cp.conv2d(x, packed, plan, co)
torch.cuda.synchronize()
