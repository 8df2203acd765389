"""ctypes binding of the C ABI declared in include/sg_ops.h.

The shared library is built in-tree by `python -m ga_gan_amd.build` (or
`__graft_entry__.build()`).  There is no fallback: a missing or stale library is
an ImportError, a failing call is a RuntimeError carrying `sg_last_error()`.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SG_OPS_LIBRARY') or os.path.join(_HERE, 'libsgops.so')   # override: kernel experiments

SYMBOLS = ('sg_version', 'sg_last_error', 'sg_init', 'sg_upfirdn2d_out_size', 'sg_upfirdn2d', 'sg_bias_act',
           'sg_mbstd_forward', 'sg_mbstd_backward', 'sg_conv2d', 'sg_conv2d_wgrad', 'sg_upfirdn2d_epilogue',
           'sg_epilogue_backward', 'sg_conv2d_reduce_rows', 'sg_conv2d_kernel', 'sg_conv2d_wgrad_kernel',
           'sg_adam_step', 'sg_lerp_multi', 'sg_images_to_float', 'sg_conv2d_reduced', 'sg_conv2d_supports_mask', 'sg_conv2d_wgrad_supports_mask',
           'sg_conv2d_workspace_bytes', 'sg_conv2d_wgrad_workspace_bytes', 'sg_conv_up2_fir', 'sg_conv_up2_fir_supported',
           'sg_upfirdn2d_add', 'sg_upfirdn2d_add_supported', 'sg_upfirdn2d_kernel', 'sg_planes_segments', 'sg_scale_planes', 'sg_dot_planes',
           'sg_styles', 'sg_styles_backward', 'sg_pack_weights',
           'sg_conv2d_prepared_weight_bytes', 'sg_conv2d_prepare_weights', 'sg_conv2d_supports_addend')

DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}

i32, i64, f32, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class Upfirdn2dParams(ctypes.Structure):
    _fields_ = [('x', vp), ('f', vp), ('y', vp), ('dtype', i32),
                ('up_x', i32), ('up_y', i32), ('down_x', i32), ('down_y', i32),
                ('pad_x0', i32), ('pad_y0', i32), ('flip', i32), ('gain', f32),
                ('in_w', i32), ('in_h', i32), ('in_c', i32), ('in_n', i32),
                ('in_stride_w', i64), ('in_stride_h', i64), ('in_stride_c', i64), ('in_stride_n', i64),
                ('f_w', i32), ('f_h', i32), ('f_stride_w', i64), ('f_stride_h', i64),
                ('out_w', i32), ('out_h', i32),
                ('out_stride_w', i64), ('out_stride_h', i64), ('out_stride_c', i64), ('out_stride_n', i64)]


class BiasActParams(ctypes.Structure):
    _fields_ = [('x', vp), ('b', vp), ('xref', vp), ('yref', vp), ('dy', vp), ('y', vp),
                ('dtype', i32), ('grad', i32), ('act', i32),
                ('alpha', f32), ('gain', f32), ('clamp', f32),
                ('size_x', i64), ('size_b', i32), ('step_b', i64)]


class MbstdParams(ctypes.Structure):
    _fields_ = [('x', vp), ('dy', vp), ('out', vp), ('dtype', i32),
                ('n', i32), ('c', i32), ('hw', i32), ('g', i32), ('f', i32)]


_lib = None


def lib():
    """The loaded library (loaded once per process)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError(f'{LIB_PATH} is missing: build the HIP library first (python -m ga_gan_amd.build)')
        handle = ctypes.CDLL(LIB_PATH)
        missing = [s for s in SYMBOLS if not hasattr(handle, s)]
        if missing:
            raise ImportError(f'{LIB_PATH} does not export {missing}: rebuild it')
        handle.sg_last_error.restype = ctypes.c_char_p
        handle.sg_version.restype = ctypes.c_int
        handle.sg_init.restype = ctypes.c_int
        handle.sg_upfirdn2d_out_size.argtypes = [ctypes.c_int] * 6
        for name, struct in (('sg_upfirdn2d', Upfirdn2dParams), ('sg_bias_act', BiasActParams),
                             ('sg_mbstd_forward', MbstdParams), ('sg_mbstd_backward', MbstdParams)):
            fn = getattr(handle, name)
            fn.argtypes = [ctypes.POINTER(struct), vp]
            fn.restype = ctypes.c_int
        handle.sg_upfirdn2d_epilogue.argtypes = [ctypes.POINTER(Upfirdn2dParams), vp, vp]
        handle.sg_upfirdn2d_epilogue.restype = ctypes.c_int
        handle.sg_upfirdn2d_add.argtypes = [ctypes.POINTER(Upfirdn2dParams), vp, vp]
        handle.sg_upfirdn2d_add.restype = ctypes.c_int
        handle.sg_upfirdn2d_add_supported.argtypes = [ctypes.POINTER(Upfirdn2dParams)]
        handle.sg_upfirdn2d_add_supported.restype = ctypes.c_int
        handle.sg_upfirdn2d_kernel.argtypes = [ctypes.POINTER(Upfirdn2dParams), vp]
        handle.sg_upfirdn2d_kernel.restype = ctypes.c_char_p
        handle.sg_planes_segments.argtypes = [ctypes.c_int, i64]
        handle.sg_planes_segments.restype = ctypes.c_int
        handle.sg_scale_planes.argtypes = [vp, vp, vp, ctypes.c_int, i64, i64, vp]
        handle.sg_scale_planes.restype = ctypes.c_int
        handle.sg_dot_planes.argtypes = [vp, vp, vp, ctypes.c_int, i64, i64, vp]
        handle.sg_dot_planes.restype = ctypes.c_int
        for name in ('sg_conv2d', 'sg_conv2d_wgrad', 'sg_epilogue_backward', 'sg_adam_step', 'sg_lerp_multi', 'sg_images_to_float', 'sg_conv2d_reduced', 'sg_conv2d_supports_mask', 'sg_conv2d_wgrad_supports_mask'):     # structs declared in the plugin modules
            getattr(handle, name).argtypes = [vp, vp]
            getattr(handle, name).restype = ctypes.c_int
        for name in ('sg_styles', 'sg_styles_backward', 'sg_pack_weights',
           'sg_conv2d_prepared_weight_bytes', 'sg_conv2d_prepare_weights', 'sg_conv2d_supports_addend'):
            getattr(handle, name).argtypes = [vp, vp]
            getattr(handle, name).restype = ctypes.c_int
        handle.sg_conv2d_reduced.argtypes = [vp, vp, vp]
        handle.sg_conv2d_reduced.restype = ctypes.c_int
        handle.sg_conv_up2_fir.argtypes = [vp, vp]
        handle.sg_conv_up2_fir.restype = ctypes.c_int
        for name in ('sg_conv2d_supports_mask', 'sg_conv2d_wgrad_supports_mask', 'sg_conv_up2_fir_supported', 'sg_conv2d_supports_addend'):
            getattr(handle, name).argtypes = [vp]
            getattr(handle, name).restype = ctypes.c_int
        handle.sg_conv2d_prepare_weights.argtypes = [vp, vp, ctypes.c_size_t, vp]
        handle.sg_conv2d_prepare_weights.restype = ctypes.c_int
        for name in ('sg_conv2d_workspace_bytes', 'sg_conv2d_wgrad_workspace_bytes', 'sg_conv2d_prepared_weight_bytes'):
            getattr(handle, name).argtypes = [vp]
            getattr(handle, name).restype = ctypes.c_size_t
        handle.sg_conv2d_reduce_rows.argtypes = [vp]
        handle.sg_conv2d_reduce_rows.restype = ctypes.c_int
        handle.sg_conv2d_kernel.argtypes = [vp]
        handle.sg_conv2d_kernel.restype = ctypes.c_char_p
        handle.sg_conv2d_wgrad_kernel.argtypes = [vp]
        handle.sg_conv2d_wgrad_kernel.restype = ctypes.c_char_p
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what} failed ({rc}): {lib().sg_last_error().decode()}')


def dtype_code(t):
    try:
        return DTYPES[t.dtype]
    except KeyError:
        raise RuntimeError(f'unsupported dtype {t.dtype}') from None


def stream_of(t):
    """The HIP stream torch is currently enqueueing on for t's device."""
    return vp(torch.cuda.current_stream(t.device).cuda_stream)


def ptr(t):
    return vp(t.data_ptr()) if t is not None and t.numel() > 0 else vp(None)


# ---- launching and per-launch timing -------------------------------------------------
_profile = None


def profile_begin():
    """Start recording one (label, algorithmic bytes, HIP event pair) entry per kernel launch.
    The events are recorded on the stream the kernel is enqueued on."""
    global _profile
    _profile = []


def profiling():
    """True while launches are being recorded."""
    return _profile is not None


def profile_end():
    """Stop recording; returns [(label, bytes, milliseconds, flops)] after synchronising."""
    global _profile
    records, _profile = _profile or [], None
    torch.cuda.synchronize()
    return [(label, nbytes, start.elapsed_time(end), flops) for label, nbytes, start, end, flops in records]


def launch(fn_name, params, ref, label=None, nbytes=0, flops=0.0):
    """Call entry point `fn_name(params*, stream)` on ref's device and current stream."""
    fn = getattr(lib(), fn_name)
    with torch.cuda.device(ref.device):
        if _profile is None:
            rc = fn(ctypes.byref(params), stream_of(ref))
        else:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            rc = fn(ctypes.byref(params), stream_of(ref))
            end.record()
            _profile.append((label or fn_name, nbytes, start, end, flops))
    check(rc, fn_name)


def launch_raw(call, ref, what, label, nbytes=0, flops=0.0):
    """Like `launch` for entry points with a different argument list: `call(stream)` returns the status."""
    with torch.cuda.device(ref.device):
        if _profile is None:
            rc = call(stream_of(ref))
        else:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            rc = call(stream_of(ref))
            end.record()
            _profile.append((label, nbytes, start, end, flops))
    check(rc, what)


def nbytes(*tensors):
    return sum(t.numel() * t.element_size() for t in tensors if t is not None)
