"""Time the augmentation pipe alone (forward + backward, and forward only) at the bench batch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ga_gan_amd.training.training_loop import make_augment_pipe
from ga_gan_amd.torch_utils.ops import grid_sample_gradfix
grid_sample_gradfix.enabled = True
pipe = make_augment_pipe('bgc').cuda()
pipe.p.fill_(0.6)
x = torch.randn(32, 3, 256, 256, device='cuda', requires_grad=True)
def run():
    y = pipe(x)
    y.square().sum().backward()
for _ in range(3): run()
torch.cuda.synchronize(); t = time.time()
for _ in range(10): run()
torch.cuda.synchronize(); print('pipe fwd+bwd ms', (time.time() - t) * 100)
with torch.no_grad():
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): pipe(x)
    torch.cuda.synchronize(); print('pipe fwd ms', (time.time() - t) * 100)
