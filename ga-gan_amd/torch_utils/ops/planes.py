"""Per-plane scale of a [N, C, H, W] tensor by float32 factors [N, C] and its gradient (a per-plane dot product), as a
pair of autograd functions that is closed under differentiation: each one's backward consists of calls of the two.

They stand for the broadcast multiplies of the reference's `modulated_conv2d` (training/networks.py:642 `x * styles`,
:648-650 `x * dcoefs`) in the layers that are evaluated from differentiable primitives (the regularisation passes):
autograd's own derivative of a broadcast multiply is a full-size product followed by a reduction, two passes where
`sg_dot_planes` makes one that writes nothing of size.  The factors stay float32 (the reference rounds them to the
tensor's dtype first)."""
import ctypes

import torch

from ... import _lib


def _eligible(x):
    per = 16 // x.element_size()
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.float16, torch.bfloat16)
            and (x.shape[2] * x.shape[3]) % per == 0 and x.numel() > 0)


def _scale(x, s):
    x = x.contiguous()
    y = torch.empty_like(x)
    n, c, h, w = x.shape
    fn = _lib.lib().sg_scale_planes
    _lib.launch_raw(lambda stream: fn(_lib.ptr(x), _lib.ptr(s), _lib.ptr(y), _lib.dtype_code(x), n * c, h * w, stream), x,
                    'sg_scale_planes', 'scale_planes_kernel', _lib.nbytes(x, y))
    return y


def _dot(a, b):
    a, b = a.contiguous(), b.contiguous()
    n, c, h, w = a.shape
    segs = _lib.lib().sg_planes_segments(_lib.dtype_code(a), h * w)
    part = torch.empty([n, c, segs], dtype=torch.float32, device=a.device)
    fn = _lib.lib().sg_dot_planes
    _lib.launch_raw(lambda stream: fn(_lib.ptr(a), _lib.ptr(b), _lib.ptr(part), _lib.dtype_code(a), n * c, h * w, stream), a,
                    'sg_dot_planes', 'dot_planes_kernel', _lib.nbytes(a, b))
    return part.sum(dim=2) if segs > 1 else part[:, :, 0]


class _ScalePlanes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(x, s)
        return _scale(x, s.contiguous())

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        gx = _ScalePlanes.apply(g, s) if ctx.needs_input_grad[0] else None
        gs = _DotPlanes.apply(g, x) if ctx.needs_input_grad[1] else None
        return gx, gs


class _DotPlanes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _dot(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.to(torch.float32)
        ga = _ScalePlanes.apply(b, g) if ctx.needs_input_grad[0] else None
        gb = _ScalePlanes.apply(a, g) if ctx.needs_input_grad[1] else None
        return ga, gb


def scale_planes(x, s):
    """x[N, C, H, W] * s[N, C] (float32), in x's dtype."""
    assert s.dtype == torch.float32 and tuple(s.shape) == tuple(x.shape[:2])
    if not _eligible(x):
        return x * s.to(x.dtype).reshape(x.shape[0], -1, 1, 1)
    return _ScalePlanes.apply(x, s)


def dot_planes(a, b):
    """sum over H, W of a * b for a, b [N, C, H, W] of one dtype -> float32 [N, C]."""
    assert a.shape == b.shape and a.dtype == b.dtype
    if not _eligible(a):
        return (a * b).sum(dim=[2, 3]).to(torch.float32)
    return _DotPlanes.apply(a, b)
