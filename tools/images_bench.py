"""Time `sg_images_to_float` at the batch sizes of the two bench configurations (HIP events on the launch stream)."""
import sys

import torch

sys.path.insert(0, __file__.rsplit('/tools/', 1)[0])
from ga_gan_amd import images_plugin  # noqa: E402

dev = 'cuda'
for count, n, res, dtype in ((4096, 32, 256, torch.float32), (256, 16, 1024, torch.float32), (256, 32, 1024, torch.float32),
                             (256, 32, 1024, torch.bfloat16)):
    images = torch.randint(0, 256, [count, 3, res, res], dtype=torch.uint8, device=dev)
    index = torch.randint(0, count, [n], device=dev)
    flip = torch.randint(0, 2, [n], device=dev).to(torch.uint8)
    for _ in range(5):
        out = images_plugin.images_to_float(images, index, flip, dtype=dtype)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    start.record()
    for _ in range(reps):
        out = images_plugin.images_to_float(images, index, flip, dtype=dtype)
    end.record()
    torch.cuda.synchronize()
    us = start.elapsed_time(end) / reps * 1e3
    nbytes = out.numel() * (1 + out.element_size())
    # the torch composition it replaces
    start.record()
    for _ in range(reps):
        ref = images[index]
        ref = torch.where(flip.bool()[:, None, None, None], ref.flip(3), ref).to(torch.float32) / 127.5 - 1
    end.record()
    torch.cuda.synchronize()
    us_torch = start.elapsed_time(end) / reps * 1e3
    print(f'{n} x 3 x {res}^2 -> {str(dtype).split(".")[1]}: {us:.1f} us, {nbytes / us / 1e3:.0f} GB/s ({nbytes / us / 1e3 / 8000:.1%} of 8 TB/s); torch ops {us_torch:.1f} us')
