"""Native entry point `upfirdn2d(...)` with the argument list of the reference's
`upfirdn2d_plugin` (torch_utils/ops/upfirdn2d.cpp:16, bound at :98-101), executed by
the gfx950 kernel behind `sg_upfirdn2d`.  Errors surface as RuntimeError, as the
reference's TORCH_CHECKs do (upfirdn2d.cpp:19-36)."""
import ctypes
import os

import torch

from . import _lib

_DETAIL = bool(os.environ.get('SG_PROFILE_DETAIL'))   # geometry in the profiling labels


def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
    if not x.is_cuda:
        raise RuntimeError('x must reside on CUDA device')
    if f.device != x.device:
        raise RuntimeError('f must reside on the same device as x')
    if f.dtype != torch.float32:
        raise RuntimeError('f must be float32')
    if x.dim() != 4:
        raise RuntimeError('x must be rank 4')
    if f.dim() != 2:
        raise RuntimeError('f must be rank 2')
    if f.shape[0] < 1 or f.shape[1] < 1:
        raise RuntimeError('f must be at least 1x1')
    if upx < 1 or upy < 1:
        raise RuntimeError('upsampling factor must be at least 1')
    if downx < 1 or downy < 1:
        raise RuntimeError('downsampling factor must be at least 1')
    if x.numel() > 2 ** 31 - 1:
        raise RuntimeError('x is too large')

    n, c, h, w = x.shape
    out_w = (w * upx + padx0 + padx1 - f.shape[1] + downx) // downx
    out_h = (h * upy + pady0 + pady1 - f.shape[0] + downy) // downy
    if out_w < 1 or out_h < 1:
        raise RuntimeError('output must be at least 1x1')
    channels_last = c > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty([n, c, out_h, out_w], dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    if y.numel() > 2 ** 31 - 1:
        raise RuntimeError('output is too large')
    if y.numel() == 0:
        return y

    p = _lib.Upfirdn2dParams(
        x=_lib.ptr(x), f=_lib.ptr(f), y=_lib.ptr(y), dtype=_lib.dtype_code(x),
        up_x=upx, up_y=upy, down_x=downx, down_y=downy, pad_x0=padx0, pad_y0=pady0, flip=1 if flip else 0, gain=gain,
        in_w=w, in_h=h, in_c=c, in_n=n,
        in_stride_w=x.stride(3), in_stride_h=x.stride(2), in_stride_c=x.stride(1), in_stride_n=x.stride(0),
        f_w=f.shape[1], f_h=f.shape[0], f_stride_w=f.stride(1), f_stride_h=f.stride(0),
        out_w=out_w, out_h=out_h,
        out_stride_w=y.stride(3), out_stride_h=y.stride(2), out_stride_c=y.stride(1), out_stride_n=y.stride(0))
    label = ''
    if _lib.profiling():     # the symbol a profiler shows for this launch
        label = _lib.lib().sg_upfirdn2d_kernel(ctypes.byref(p), None).decode()
        if label == 'upfirdn2d_tiled_kernel':
            label += f'<up{upx} down{downx}>'
    if _DETAIL and label:
        label += f' {c}x{h}x{w} n{n} {str(x.dtype)[6:]}'
    _lib.launch('sg_upfirdn2d', p, x, label, _lib.nbytes(x, y))
    return y


def upfirdn2d_add(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, addend):
    """round(upfirdn2d(x, ...)) + addend in one pass (`sg_upfirdn2d_add`), or None when no kernel covers the geometry
    (the caller then filters and adds in two steps).  `addend`: a dense tensor of the result's shape and dtype."""
    if not (x.is_cuda and addend.is_cuda and f.is_cuda) or x.dim() != 4 or f.dim() != 2 or f.dtype != torch.float32:
        return None
    n, c, h, w = x.shape
    out_w = (w * upx + padx0 + padx1 - f.shape[1] + downx) // downx
    out_h = (h * upy + pady0 + pady1 - f.shape[0] + downy) // downy
    if tuple(addend.shape) != (n, c, out_h, out_w) or addend.dtype != x.dtype or not addend.is_contiguous() or x.numel() == 0:
        return None
    if x.numel() > 2 ** 31 - 1 or addend.numel() > 2 ** 31 - 1:
        return None
    y = torch.empty_like(addend)
    p = _lib.Upfirdn2dParams(
        x=_lib.ptr(x), f=_lib.ptr(f), y=_lib.ptr(y), dtype=_lib.dtype_code(x),
        up_x=upx, up_y=upy, down_x=downx, down_y=downy, pad_x0=padx0, pad_y0=pady0, flip=1 if flip else 0, gain=gain,
        in_w=w, in_h=h, in_c=c, in_n=n,
        in_stride_w=x.stride(3), in_stride_h=x.stride(2), in_stride_c=x.stride(1), in_stride_n=x.stride(0),
        f_w=f.shape[1], f_h=f.shape[0], f_stride_w=f.stride(1), f_stride_h=f.stride(0),
        out_w=out_w, out_h=out_h,
        out_stride_w=y.stride(3), out_stride_h=y.stride(2), out_stride_c=y.stride(1), out_stride_n=y.stride(0))
    if not _lib.lib().sg_upfirdn2d_add_supported(ctypes.byref(p)):
        return None
    label = 'up2_stream_kernel'
    if _DETAIL:
        label += f' {c}x{h}x{w} n{n} {str(x.dtype)[6:]}'
    fn = _lib.lib().sg_upfirdn2d_add
    _lib.launch_raw(lambda stream: fn(ctypes.byref(p), _lib.ptr(addend), stream), x, 'sg_upfirdn2d_add', label,
                    _lib.nbytes(x, y, addend))
    return y
