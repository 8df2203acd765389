"""a * b + c with broadcasting and cheap gradients (reference: torch_utils/ops/fma.py)."""
import torch

from . import conv2d_gradfix


def fma(a, b, c):
    return _Fma.apply(a, b, c)


def _reduce_to(x, shape):
    """Sum x over the dimensions that were broadcast from `shape`."""
    extra = x.ndim - len(shape)
    dims = [i for i in range(x.ndim) if x.shape[i] > 1 and (i < extra or shape[i - extra] == 1)]
    if dims:
        x = x.sum(dim=dims, keepdim=True)
    if extra:
        x = x.reshape(-1, *x.shape[extra + 1:])
    assert x.shape == shape
    return x


class _Fma(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = _reduce_to(dout * b, a.shape) if ctx.needs_input_grad[0] else None
        db = _reduce_to(dout * a, b.shape) if ctx.needs_input_grad[1] else None
        # c is the noise term of a synthesis layer (noise buffer x learned strength): like a weight gradient, its gradient
        # is not wanted inside conv2d_gradfix.no_weight_gradients()
        dc = _reduce_to(dout, ctx.c_shape) if ctx.needs_input_grad[2] and not conv2d_gradfix.parameter_gradients_disabled else None
        return da, db, dc
