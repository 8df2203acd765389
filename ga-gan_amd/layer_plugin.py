"""Bindings of the layer-epilogue entry points (include/sg_ops.h): `sg_upfirdn2d_epilogue`
(blur with the demodulation / noise / bias / activation epilogue in the same pass) and
`sg_epilogue_backward` (gradient through that epilogue plus the bias, scale and noise-strength
reductions in one pass over dy and y)."""
import ctypes

import torch

from . import _lib
from .conv2d_plugin import ACT_INDEX


class Epilogue(ctypes.Structure):
    _fields_ = [('scale', ctypes.c_void_p), ('noise', ctypes.c_void_p), ('noise_gain', ctypes.c_void_p),
                ('noise_stride_n', ctypes.c_int64), ('bias', ctypes.c_void_p), ('act', ctypes.c_int32),
                ('alpha', ctypes.c_float), ('gain', ctypes.c_float), ('clamp', ctypes.c_float)]


class EpilogueBwdParams(ctypes.Structure):
    _fields_ = [('dy', ctypes.c_void_p), ('y', ctypes.c_void_p), ('dz', ctypes.c_void_p), ('ep', Epilogue),
                ('dbias', ctypes.c_void_p), ('dscale', ctypes.c_void_p), ('dnoise_gain', ctypes.c_void_p),
                ('dtype', ctypes.c_int32), ('n', ctypes.c_int32), ('c', ctypes.c_int32), ('h', ctypes.c_int32),
                ('w', ctypes.c_int32)]


def make_epilogue(n, c, oh, ow, dtype, scale=None, noise=None, noise_gain=None, bias=None, act='linear', alpha=0.2,
                  gain=1.0, clamp=-1.0):
    """Validated `sg_epilogue` for an output of shape [n, c, oh, ow]."""
    if scale is not None and (scale.dtype != torch.float32 or tuple(scale.shape) != (n, c) or not scale.is_contiguous()):
        raise RuntimeError(f'scale must be a dense float32 tensor of shape {(n, c)}')
    if bias is not None and (bias.dtype != dtype or bias.numel() != c or not bias.is_contiguous()):
        raise RuntimeError('bias must be dense, of the tensor\'s dtype, with one entry per channel')
    stride = 0
    if noise is not None:
        if noise.dtype != torch.float32 or not noise.is_contiguous() or tuple(noise.shape[-2:]) != (oh, ow):
            raise RuntimeError('noise must be dense float32 [.., oh, ow]')
        if noise.numel() == n * oh * ow and n > 1:
            stride = oh * ow
        elif noise.numel() != oh * ow:
            raise RuntimeError('noise must hold one plane, or one plane per sample')
    if noise_gain is not None and (noise_gain.dtype != torch.float32 or noise_gain.numel() != 1):
        raise RuntimeError('noise_gain must be a float32 scalar tensor')
    return Epilogue(scale=_lib.ptr(scale), noise=_lib.ptr(noise), noise_gain=_lib.ptr(noise_gain), noise_stride_n=stride,
                    bias=_lib.ptr(bias), act=ACT_INDEX[act], alpha=alpha, gain=gain, clamp=clamp)


def blur_epilogue(x, f, padding, flip, filter_gain, **epilogue):
    """upfirdn2d(x, f, up=1, down=1, padding=[x0, x1, y0, y1], gain=filter_gain) followed by the epilogue, one kernel."""
    if not x.is_cuda or x.dim() != 4 or not x.is_contiguous():
        raise RuntimeError('x must be a dense NCHW tensor on the GPU')
    if f.dtype != torch.float32 or f.dim() != 2 or f.shape[0] > 4 or f.shape[1] > 4:
        raise RuntimeError('f must be a float32 filter of at most 4x4 taps')
    n, c, h, w = x.shape
    px0, px1, py0, py1 = padding
    ow, oh = w + px0 + px1 - f.shape[1] + 1, h + py0 + py1 - f.shape[0] + 1
    if ow < 1 or oh < 1:
        raise RuntimeError('output must be at least 1x1')
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device)
    if y.numel() == 0:
        return y
    ep = make_epilogue(n, c, oh, ow, x.dtype, **epilogue)
    p = _lib.Upfirdn2dParams(
        x=_lib.ptr(x), f=_lib.ptr(f), y=_lib.ptr(y), dtype=_lib.dtype_code(x), up_x=1, up_y=1, down_x=1, down_y=1,
        pad_x0=px0, pad_y0=py0, flip=1 if flip else 0, gain=filter_gain, in_w=w, in_h=h, in_c=c, in_n=n,
        in_stride_w=x.stride(3), in_stride_h=x.stride(2), in_stride_c=x.stride(1), in_stride_n=x.stride(0),
        f_w=f.shape[1], f_h=f.shape[0], f_stride_w=f.stride(1), f_stride_h=f.stride(0), out_w=ow, out_h=oh,
        out_stride_w=y.stride(3), out_stride_h=y.stride(2), out_stride_c=y.stride(1), out_stride_n=y.stride(0))
    _lib.launch_raw(lambda stream: _lib.lib().sg_upfirdn2d_epilogue(ctypes.byref(p), ctypes.byref(ep), stream), x,
                    'sg_upfirdn2d_epilogue', 'upfirdn2d_blur4_kernel', _lib.nbytes(x, y))
    return y


def epilogue_backward(dy, y, want_dbias, want_dscale, want_dnoise_gain, want_dz=True, **epilogue):
    """-> (dz | None, dbias [c] | None, dscale [n, c] | None, dnoise_gain [] | None); all reductions float32."""
    if not dy.is_cuda or dy.shape != y.shape or dy.dtype != y.dtype or y.dim() != 4:
        raise RuntimeError('dy and y must be GPU tensors of the same shape and dtype')
    dy, y = dy.contiguous(), y.contiguous()
    n, c, h, w = y.shape
    ep = make_epilogue(n, c, h, w, y.dtype, **epilogue)
    dz = torch.empty_like(y) if want_dz else None
    dbias = torch.zeros([c], dtype=torch.float32, device=y.device) if want_dbias else None
    dscale = torch.zeros([n, c], dtype=torch.float32, device=y.device) if want_dscale else None
    dng = torch.zeros([], dtype=torch.float32, device=y.device) if want_dnoise_gain else None
    if y.numel():
        p = EpilogueBwdParams(dy=_lib.ptr(dy), y=_lib.ptr(y), dz=_lib.ptr(dz), ep=ep, dbias=_lib.ptr(dbias),
                              dscale=_lib.ptr(dscale), dnoise_gain=_lib.ptr(dng), dtype=_lib.dtype_code(y), n=n, c=c, h=h, w=w)
        _lib.launch('sg_epilogue_backward', p, y, 'epilogue_bwd_kernel', _lib.nbytes(dy, y, dz))
    return dz, dbias, dscale, dng
