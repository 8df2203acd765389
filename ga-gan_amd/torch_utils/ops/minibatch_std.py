"""Minibatch standard deviation as one GPU op (the reference computes it with seven
tensor ops inside `MinibatchStdLayer.forward`, training/networks.py:1284-1301).

First-order backward is a kernel; when a higher-order graph is being recorded
(R1 regularisation differentiates through the discriminator's backward) the
backward re-expresses the statistic with differentiable tensor ops instead.
"""
import torch

from ... import _lib
from .. import misc


MAX_KERNEL_GROUP = 64   # samples per group the kernels hold (sg_mbstd rejects larger groups)


def _composite(x, group_size, num_channels):
    n, c, h, w = x.shape
    g = min(group_size, n) if group_size is not None else n
    f = num_channels
    y = x.reshape(g, -1, f, c // f, h, w)
    y = y - y.mean(dim=0)
    y = (y.square().mean(dim=0) + 1e-8).sqrt()
    y = y.mean(dim=[2, 3, 4]).reshape(-1, f, 1, 1).repeat(g, 1, h, w)
    return torch.cat([x, y], dim=1)


def _call(fn_name, x, dy, out, g, f):
    n, c, h, w = x.shape
    p = _lib.MbstdParams(x=_lib.ptr(x), dy=_lib.ptr(dy), out=_lib.ptr(out), dtype=_lib.dtype_code(x),
                         n=n, c=c, hw=h * w, g=g, f=f)
    _lib.launch(fn_name, p, x, 'mbstd_kernel', _lib.nbytes(x, dy, out))


class _MinibatchStd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, f):
        n, c, h, w = x.shape
        y = torch.empty([n, c + f, h, w], dtype=x.dtype, device=x.device)
        _call('sg_mbstd_forward', x, None, y, g, f)
        ctx.save_for_backward(x)
        ctx.g, ctx.f = g, f
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        if torch.is_grad_enabled():  # double backward requested: stay differentiable
            with torch.enable_grad():
                xx = x.detach().requires_grad_(True) if not x.requires_grad else x
                y = _composite(xx, ctx.g, ctx.f)
                dx, = torch.autograd.grad(y, xx, dy, create_graph=True)
            return dx, None, None
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _call('sg_mbstd_backward', x, dy, dx, ctx.g, ctx.f)
        return dx, None, None


def minibatch_std(x, group_size, num_channels=1):
    """x[N, C, H, W] -> [N, C + num_channels, H, W]."""
    misc.require_gpu(x, 'minibatch_std')
    n, c = x.shape[:2]
    g = min(int(group_size), n) if group_size is not None else n
    if n % g != 0 or c % num_channels != 0:
        raise RuntimeError(f'minibatch_std: batch {n} / channels {c} not divisible by group {g} / sets {num_channels}')
    if g > MAX_KERNEL_GROUP:     # (whole-batch groups of a large batch: the tensor ops of the reference)
        return _composite(x, g, num_channels)
    # made contiguous here, inside the caller's graph: the copy the function would make itself is a new leaf, and
    # the second-order term through the statistic would end at it
    return _MinibatchStd.apply(x.contiguous(), g, num_channels)
