"""Times strided 3x3 convolutions (the discriminator's down-sampling layers after the filter), bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_gan_amd import conv2d_plugin as cp
for n, ci, co, h in [(64, 512, 512, 33), (32, 512, 512, 33), (64, 256, 512, 65), (64, 512, 512, 17)]:
    x = torch.randn([n, ci, h, h + 7], device='cuda').to(torch.bfloat16)[:, :, :, :h].contiguous() if False else torch.randn([n, ci, h, (h + 7) // 8 * 8], device='cuda').to(torch.bfloat16)
    w = (torch.randn([co, ci, 3, 3], device='cuda') / 50).to(torch.bfloat16)
    plan = cp.plan_conv(3, 3, stride=2, padding=0); packed = plan.pack(w)
    out_hw = ((h - 3) // 2 + 1, (h - 3) // 2 + 1)
    run = lambda: cp.conv2d(x, packed, plan, co, out_hw=out_hw)
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20
    fl = 2.0 * n * ci * co * 9 * out_hw[0] * out_hw[1]
    print(f'{cp.kernel_name(x, plan, co, out_hw=out_hw) if hasattr(cp, "kernel_name") else ""} n{n} {ci}->{co} {h}->{out_hw[0]}: {t:.3f} ms {fl / t / 1e9:.0f} TF/s', flush=True)
