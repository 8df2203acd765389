"""Native entry point `bias_act(...)` with the argument list of the reference's
`bias_act_plugin` (torch_utils/ops/bias_act.cpp:32, bound at :94-97), executed by the
gfx950 kernel behind `sg_bias_act`.  "Absent" tensors are passed as empty tensors, as
the reference's Python wrapper does (torch_utils/ops/bias_act.py:67,186-196)."""
import torch

from . import _lib


def _same_layout(a, b):
    return a.dim() == b.dim() and all(a.size(i) < 2 or a.stride(i) == b.stride(i) for i in range(a.dim()))


def _is_permuted_dense(t):
    order = sorted(range(t.dim()), key=lambda i: (-t.stride(i), -t.size(i)))
    expect = 1
    for i in reversed(order):
        if t.size(i) != 1 and t.stride(i) != expect:
            return False
        expect *= t.size(i)
    return True


def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
    if not x.is_cuda:
        raise RuntimeError('x must reside on CUDA device')
    for name, t in (('xref', xref), ('yref', yref), ('dy', dy)):
        if t.numel() != 0 and (t.shape != x.shape or t.dtype != x.dtype or t.device != x.device):
            raise RuntimeError(f'{name} must have the same shape, dtype, and device as x')
        if t.numel() != 0 and not _same_layout(t, x):
            raise RuntimeError(f'{name} must have the same layout as x')
    if b.numel() != 0 and (b.dtype != x.dtype or b.device != x.device):
        raise RuntimeError('b must have the same dtype and device as x')
    if x.numel() > 2 ** 31 - 1:
        raise RuntimeError('x is too large')
    if b.dim() != 1:
        raise RuntimeError('b must have rank 1')
    if b.numel() != 0 and not (0 <= dim < x.dim()):
        raise RuntimeError('dim is out of bounds')
    if b.numel() != 0 and b.numel() != x.size(dim):
        raise RuntimeError('b has wrong number of elements')
    if grad < 0:
        raise RuntimeError('grad must be non-negative')
    if not _is_permuted_dense(x):
        raise RuntimeError('x must be non-overlapping and dense')
    if not b.is_contiguous():
        raise RuntimeError('b must be contiguous')

    y = torch.empty_like(x)  # preserves the dense layout of x
    if not _same_layout(y, x):
        raise RuntimeError('y must have the same layout as x')
    if x.numel() == 0:
        return y
    p = _lib.BiasActParams(
        x=_lib.ptr(x), b=_lib.ptr(b), xref=_lib.ptr(xref), yref=_lib.ptr(yref), dy=_lib.ptr(dy), y=_lib.ptr(y),
        dtype=_lib.dtype_code(x), grad=grad, act=act, alpha=alpha, gain=gain, clamp=clamp,
        size_x=x.numel(), size_b=b.numel(), step_b=x.stride(dim) if b.numel() else 1)
    _lib.launch('sg_bias_act', p, x, f'bias_act_kernel<act{act}>', _lib.nbytes(x, xref, yref, dy, y))
    return y
