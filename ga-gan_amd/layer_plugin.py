"""Bindings of the layer-epilogue entry points (include/sg_ops.h): `sg_upfirdn2d_epilogue`
(blur with the demodulation / noise / bias / activation epilogue in the same pass),
`sg_epilogue_backward` (gradient through that epilogue plus the bias, scale and noise-strength
reductions in one pass over dy and y) and `sg_conv_up2_fir` (a whole up-sampling layer -
transposed convolution, resampling filter, epilogue - in one pass)."""
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
                ('w', ctypes.c_int32), ('dnoise_slots', ctypes.c_int32)]


NOISE_SLOTS = 256     # addresses the noise-strength gradient is accumulated over (one address takes an atomic per 14 ns)


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
    label = _lib.lib().sg_upfirdn2d_kernel(ctypes.byref(p), ctypes.byref(ep)).decode() if _lib.profiling() else ''
    _lib.launch_raw(lambda stream: _lib.lib().sg_upfirdn2d_epilogue(ctypes.byref(p), ctypes.byref(ep), stream), x,
                    'sg_upfirdn2d_epilogue', label, _lib.nbytes(x, y))
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
    dng = torch.zeros([NOISE_SLOTS], dtype=torch.float32, device=y.device) if want_dnoise_gain else None
    if y.numel():
        p = EpilogueBwdParams(dy=_lib.ptr(dy), y=_lib.ptr(y), dz=_lib.ptr(dz), ep=ep, dbias=_lib.ptr(dbias),
                              dscale=_lib.ptr(dscale), dnoise_gain=_lib.ptr(dng), dtype=_lib.dtype_code(y), n=n, c=c, h=h, w=w,
                              dnoise_slots=NOISE_SLOTS)
        _lib.launch('sg_epilogue_backward', p, y, 'epilogue_bwd_kernel', _lib.nbytes(dy, y, dz))
    return dz, dbias, dscale, (dng.sum() if dng is not None else None)


class ConvUp2FirParams(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('y', ctypes.c_void_p), ('in_scale', ctypes.c_void_p),
                ('fy4', ctypes.c_void_p), ('ep', Epilogue), ('dtype', ctypes.c_int32), ('n', ctypes.c_int32),
                ('ci', ctypes.c_int32), ('h', ctypes.c_int32), ('w_in', ctypes.c_int32), ('co', ctypes.c_int32)]


def separable_taps(f):
    """(m, fy4) for a 4 x 4 float32 filter of rank one, f = outer(fy, fx), else None: m [3, 6] maps a weight row
    (3 taps kx) to its convolution with 2 * fx (6 taps d = kx - j + 1 for filter tap fx[3 - j]) in the column order
    d = -2, 0, 2, -1, 1, 3 of `sg_conv_up2_fir`; fy4 holds 2 * fy[3 - i].  Both stay on the device.  The rank check
    reads the filter back once: callers cache the answer per filter tensor."""
    if f is None or f.dim() != 2 or tuple(f.shape) != (4, 4) or f.dtype != torch.float32:
        return None
    total = f.sum()
    fy, fx = f.sum(dim=1), f.sum(dim=0) / total
    if not bool(torch.isfinite(fx).all()) or not torch.allclose(torch.outer(fy, fx), f, rtol=1e-5, atol=1e-7):
        return None
    m = f.new_zeros([3, 6])
    for kx in range(3):
        m[kx, kx:kx + 4] = 2.0 * fx          # column d + 2 = kx + (3 - j)
    m = torch.cat([m[:, 0::2], m[:, 1::2]], dim=1).contiguous()
    return m, (2.0 * fy.flip(0)).contiguous()


def compose_up2_weights(w, m):
    """Weights of `conv_up2_fir`: every row of w [co, ci, 3, 3] (the kernel of the stride-2 transposed convolution,
    t[2a+ky, 2b+kx] += x[a, b] * w[ky, kx]) convolved with the horizontal half of the resampling filter and its share
    of the gain (`m` of `separable_taps`), laid out [row tap: ky 0, 2, 1][column tap][co][ci] in w's dtype."""
    full = torch.matmul(w.to(torch.float32), m)            # [co, ci, ky, column tap]
    rows = torch.cat([full[:, :, 0:1], full[:, :, 2:3], full[:, :, 1:2]], dim=2)
    return rows.permute(2, 3, 0, 1).contiguous().to(w.dtype)


def _up2_fir_params(x, w_fir, y, fy4, co, in_scale, ep):
    n, ci, h, w_in = x.shape
    return ConvUp2FirParams(x=_lib.ptr(x), w=_lib.ptr(w_fir), y=_lib.ptr(y), in_scale=_lib.ptr(in_scale), fy4=_lib.ptr(fy4), ep=ep,
                            dtype=_lib.dtype_code(x), n=n, ci=ci, h=h, w_in=w_in, co=co)


def conv_up2_fir_supported(x, co, act='linear', alpha=0.2, gain=1.0):
    """True when `conv_up2_fir` has a kernel for x [n, ci, h, w] -> [n, co, 2h, 2w] (16-bit, dense, aligned)."""
    if not x.is_cuda or x.dim() != 4 or not x.is_contiguous() or x.dtype not in (torch.float16, torch.bfloat16):
        return False
    ep = Epilogue(act=ACT_INDEX[act], alpha=alpha, gain=gain, clamp=-1.0)
    p = _up2_fir_params(x, x, x, x, co, None, ep)
    return bool(_lib.lib().sg_conv_up2_fir_supported(ctypes.byref(p)))


def conv_up2_fir(x, w_fir, co, fy4, in_scale=None, **epilogue):
    """epilogue(upfirdn2d(conv_transpose2d(x * in_scale, w, stride 2), f, padding 1, gain 4)) in one kernel.
    (m, fy4) = separable_taps(f); w_fir = compose_up2_weights(w, m)."""
    if not x.is_cuda or x.dim() != 4 or not x.is_contiguous():
        raise RuntimeError('x must be a dense NCHW tensor on the GPU')
    n, ci, h, w_in = x.shape
    if w_fir.dtype != x.dtype or w_fir.numel() != 18 * co * ci or not w_fir.is_contiguous():
        raise RuntimeError('composed weights must be dense [3, 6, co, ci] of x\'s dtype')
    if fy4.dtype != torch.float32 or fy4.numel() != 4 or not fy4.is_contiguous():
        raise RuntimeError('fy4 must hold four float32 values')
    if in_scale is not None and (in_scale.dtype != torch.float32 or tuple(in_scale.shape) != (n, ci) or not in_scale.is_contiguous()):
        raise RuntimeError(f'in_scale must be a dense float32 tensor of shape {(n, ci)}')
    y = torch.empty([n, co, 2 * h, 2 * w_in], dtype=x.dtype, device=x.device)
    ep = make_epilogue(n, co, 2 * h, 2 * w_in, x.dtype, **epilogue)
    p = _up2_fir_params(x, w_fir, y, fy4, co, in_scale, ep)
    flops = 2.0 * n * ci * co * 18 * h * w_in
    _lib.launch('sg_conv_up2_fir', p, x, f'conv_up2_fir_kernel<{"__hip_bfloat16" if x.dtype == torch.bfloat16 else "__half"}>',
                _lib.nbytes(x, y), flops=flops)
    return y
