"""Binding of `sg_conv2d` (include/sg_ops.h): implicit-GEMM convolution on the matrix
cores with optional per-sample input / output scales, noise, bias, leaky-ReLU, gain and
clamp fused in.  A `Plan` describes the geometry (plain, strided or transposed in
polyphase form) and how the weight tensor is packed for it."""
import ctypes
import os

import torch

from . import _lib
from .torch_utils import memo


class Phase(ctypes.Structure):
    _fields_ = [('oy0', ctypes.c_int32), ('ox0', ctypes.c_int32), ('dy0', ctypes.c_int32), ('dx0', ctypes.c_int32),
                ('nky', ctypes.c_int32), ('nkx', ctypes.c_int32), ('w_offset', ctypes.c_int64)]


class Conv2dParams(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('y', ctypes.c_void_p),
                ('in_scale', ctypes.c_void_p), ('out_scale', ctypes.c_void_p), ('bias', ctypes.c_void_p),
                ('noise', ctypes.c_void_p), ('noise_gain', ctypes.c_void_p), ('noise_stride_n', ctypes.c_int64),
                ('dtype', ctypes.c_int32),
                ('n', ctypes.c_int32), ('ci', ctypes.c_int32), ('h', ctypes.c_int32), ('w_in', ctypes.c_int32),
                ('co', ctypes.c_int32), ('oh', ctypes.c_int32), ('ow', ctypes.c_int32),
                ('in_step', ctypes.c_int32), ('out_step', ctypes.c_int32), ('num_phases', ctypes.c_int32),
                ('phase', Phase * 4),
                ('act', ctypes.c_int32), ('alpha', ctypes.c_float), ('gain', ctypes.c_float), ('clamp', ctypes.c_float),
                ('reduce_x', ctypes.c_void_p), ('reduce_out', ctypes.c_void_p),
                ('mask_y', ctypes.c_void_p), ('mask_alpha', ctypes.c_float), ('mask_clamp', ctypes.c_float),
                ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t), ('w_prepared', ctypes.c_void_p),
                ('addend', ctypes.c_void_p)]


def _attach_workspace(p, query, like):
    """Scratch memory for the float32 kernels, from torch's allocator (stream-ordered, and inside a graph capture it
    comes from the graph's own pool: the library's fallback buffer would have to grow with hipMalloc there).  The
    tensor is returned so that it outlives the launch."""
    need = int(query(ctypes.byref(p)))
    if need == 0:
        return None
    work = torch.empty([need], dtype=torch.uint8, device=like.device)
    p.workspace, p.workspace_bytes = work.data_ptr(), need
    return work


class Plan:
    """Geometry of one convolution.  `phases` = [(oy0, ox0, dy0, dx0, nky, nkx, ky_index, kx_index)] where
    ky_index / kx_index list, per kernel tap, which row / column of the ordinary weight tensor it uses."""

    def __init__(self, kind, kh, kw, in_step, out_step, phases, out_size):
        self.kind, self.kh, self.kw = kind, kh, kw
        self.in_step, self.out_step = in_step, out_step
        self.phases = phases
        self.out_size = out_size          # (h, w) -> (oh, ow)
        self.key = (kind, kh, kw, in_step, out_step, tuple((p[4], p[5], tuple(p[6]), tuple(p[7])) for p in phases))

    def pack(self, w, transposed_layout=False):
        """w [co, ci, kh, kw] (or [ci, co, kh, kw] when transposed_layout) -> flat packed
        [phase][ky][kx][co][ci] in w's dtype.  Packed once per plan for a weight tensor that is a constant."""
        return memo.memo(w, ('pack', self.key, transposed_layout), lambda: self._pack(w, transposed_layout))

    def _pack(self, w, transposed_layout):
        if transposed_layout:
            w = w.transpose(0, 1)
        if (w.is_cuda and w.dtype in _lib.DTYPES and w.dtype != torch.float64 and len(self.phases) <= 4
                and all(ph[4] <= PACK_MAX_TAPS and ph[5] <= PACK_MAX_TAPS for ph in self.phases)):
            return self._pack_kernel(w)
        parts = []
        for (_, _, _, _, nky, nkx, ky_idx, kx_idx) in self.phases:
            sel, flips = w, []
            for dim, idx in ((2, ky_idx), (3, kx_idx)):
                sel, flipped = _take(sel, dim, list(idx))
                if flipped:
                    flips.append(dim)
            if flips:
                sel = sel.flip(flips)
            parts.append(sel.permute(2, 3, 0, 1).reshape(-1))
        return torch.cat(parts) if len(parts) > 1 else parts[0].contiguous()


PACK_MAX_TAPS = 8   # SG_PACK_MAX_TAPS of include/sg_ops.h


class PackPhase(ctypes.Structure):
    _fields_ = [('nky', ctypes.c_int32), ('nkx', ctypes.c_int32), ('ky', ctypes.c_int32 * PACK_MAX_TAPS), ('kx', ctypes.c_int32 * PACK_MAX_TAPS)]


class PackParams(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('out', ctypes.c_void_p), ('dtype', ctypes.c_int32), ('co', ctypes.c_int32), ('ci', ctypes.c_int32),
                ('stride_o', ctypes.c_int64), ('stride_i', ctypes.c_int64), ('stride_y', ctypes.c_int64), ('stride_x', ctypes.c_int64),
                ('num_phases', ctypes.c_int32), ('phase', PackPhase * 4)]


def _pack_kernel(self, w):
    """`Plan._pack` as one launch of sg_pack_weights (w may be any strided view)."""
    co, ci = w.shape[0], w.shape[1]
    total = sum(ph[4] * ph[5] for ph in self.phases) * co * ci
    out = torch.empty([total], dtype=w.dtype, device=w.device)
    p = PackParams(w.data_ptr(), out.data_ptr(), _lib.dtype_code(w), co, ci, w.stride(0), w.stride(1), w.stride(2), w.stride(3),
                   len(self.phases))
    for k, (_, _, _, _, nky, nkx, ky_idx, kx_idx) in enumerate(self.phases):
        p.phase[k].nky, p.phase[k].nkx = nky, nkx
        for t, v in enumerate(ky_idx):
            p.phase[k].ky[t] = v
        for t, v in enumerate(kx_idx):
            p.phase[k].kx[t] = v
    fn = _lib.lib().sg_pack_weights
    _lib.launch_raw(lambda stream: fn(ctypes.byref(p), stream), w, 'sg_pack_weights', 'pack_weights_kernel', 2 * _lib.nbytes(out))
    return out


Plan._pack_kernel = _pack_kernel


def _take(w, dim, idx):
    """w indexed along `dim` by the arithmetic progression `idx`, as a view: (tensor, needs_flip).  Views and one
    flip kernel instead of an index tensor keep the call free of host-to-device copies (graph capture)."""
    if len(idx) == 1:
        return w.narrow(dim, idx[0], 1), False
    step = idx[1] - idx[0]
    assert step != 0 and all(b - a == step for a, b in zip(idx, idx[1:])), idx
    lo, hi = min(idx), max(idx)
    index = [slice(None)] * w.dim()
    index[dim] = slice(lo, hi + 1, abs(step))
    return w[tuple(index)], step < 0


def plan_conv(kh, kw, stride=1, padding=0):
    """Correlation (what torch's conv2d computes) with symmetric zero padding."""
    def out_size(h, w):
        return (h + 2 * padding - kh) // stride + 1, (w + 2 * padding - kw) // stride + 1
    phases = [(0, 0, -padding, -padding, kh, kw, list(range(kh)), list(range(kw)))]
    return Plan('conv', kh, kw, stride, 1, phases, out_size)


def plan_conv_transpose(kh, kw, stride=2, padding=0):
    """torch's conv_transpose2d(x, w[ci, co, kh, kw], stride, padding) in polyphase form: output row
    oy = py + stride * a gathers taps ky = r + stride * j with r = (py + padding) % stride from input rows
    a + (py + padding - r) / stride - j."""
    def axis(k, phase):
        r = (phase + padding) % stride
        taps = list(range(r, k, stride))
        count = len(taps)
        d0 = (phase + padding - r) // stride - (count - 1)
        return count, d0, taps[::-1]      # kernel tap t uses weight index taps[count-1-t]

    def out_size(h, w):
        return (h - 1) * stride + kh - 2 * padding, (w - 1) * stride + kw - 2 * padding

    phases = []
    for py in range(stride):
        for px in range(stride):
            nky, dy0, ky_idx = axis(kh, py)
            nkx, dx0, kx_idx = axis(kw, px)
            if nky and nkx:
                phases.append((py, px, dy0, dx0, nky, nkx, ky_idx, kx_idx))
    return Plan('conv_transpose', kh, kw, 1, stride, phases, out_size)


ACT_INDEX = {'linear': 1, 'lrelu': 3}
_DETAIL = bool(os.environ.get('SG_PROFILE_DETAIL'))   # profiling labels carry the geometry
_DTYPE_NAME = {torch.float32: 'float', torch.float16: '__half', torch.bfloat16: '__hip_bfloat16'}


REDUCED_MAX_PARTIALS = 1 << 20   # SG_CONV2D_REDUCED_MAX_PARTIALS of include/sg_ops.h


def conv2d(x, w_packed, plan, co, in_scale=None, out_scale=None, bias=None, noise=None, noise_gain=None,
           act='linear', alpha=0.2, gain=1.0, clamp=-1.0, out_hw=None, reduce_x=None, out=None, reduce_sum=None,
           mask_y=None, mask_alpha=1.0, mask_clamp=-1.0, workspace=None, addend=None):
    """Run `plan` on x [n, ci, h, w] (dense NCHW) with packed weights; returns y [n, co, oh, ow].
    `out_hw` overrides the plan's natural output size (pixels whose taps all fall outside x are zero).
    With `reduce_x` (y's shape and dtype) the call returns (y, r), r[n, co] = sum over pixels of
    reduce_x * (the contraction before out_scale and the epilogue), float32.
    With `out` (and, for a reduction, `reduce_sum` [n, co] float32) given, the results are written there and the
    call allocates no device memory: the reduction then runs through `sg_conv2d_reduced`, whose partial sums live
    in a fixed array inside the library (at most REDUCED_MAX_PARTIALS of them, one call at a time per device);
    a float32 call additionally takes its scratch memory from `workspace` (a uint8 tensor of at least
    `workspace_bytes(x, plan, co)` bytes) instead of allocating it.
    With `mask_y` (x's shape and dtype) x is contracted as x * m, m = 1 where mask_y > 0, else mask_alpha, and 0 where
    |mask_y| >= mask_clamp (mask_clamp < 0: unbounded): the backward of a fused activation applied to the incoming
    gradient while it is staged.  Ask `supports_mask` first."""
    if not x.is_cuda:
        raise RuntimeError('x must reside on CUDA device')
    if x.dim() != 4 or not x.is_contiguous():
        raise RuntimeError('x must be a dense NCHW tensor of rank 4')
    if w_packed.dtype != x.dtype or w_packed.device != x.device or not w_packed.is_contiguous():
        raise RuntimeError('packed weights must be dense and have the dtype and device of x')
    n, ci, h, w = x.shape
    oh, ow = plan.out_size(h, w) if out_hw is None else out_hw
    if oh < 1 or ow < 1:
        raise RuntimeError('output must be at least 1x1')
    expect = sum(ph[4] * ph[5] for ph in plan.phases) * co * ci
    if w_packed.numel() != expect:
        raise RuntimeError(f'packed weights have {w_packed.numel()} elements, the plan needs {expect}')
    for name, t, shape in (('in_scale', in_scale, (n, ci)), ('out_scale', out_scale, (n, co))):
        if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous()):
            raise RuntimeError(f'{name} must be a dense float32 tensor of shape {shape}')
    if bias is not None and (bias.dtype != x.dtype or bias.numel() != co or not bias.is_contiguous()):
        raise RuntimeError('bias must be dense, of x\'s dtype, with one entry per output channel')
    noise_stride = 0
    if noise is not None:
        if noise.dtype != torch.float32 or not noise.is_contiguous() or tuple(noise.shape[-2:]) != (oh, ow):
            raise RuntimeError('noise must be dense float32 [.., oh, ow]')
        if noise.numel() == n * oh * ow and n > 1:
            noise_stride = oh * ow
        elif noise.numel() != oh * ow:
            raise RuntimeError('noise must hold one plane, or one plane per sample')
    if noise_gain is not None and (noise_gain.dtype != torch.float32 or noise_gain.numel() != 1):
        raise RuntimeError('noise_gain must be a float32 scalar tensor')
    if out is not None:
        if tuple(out.shape) != (n, co, oh, ow) or out.dtype != x.dtype or out.device != x.device or not out.is_contiguous():
            raise RuntimeError(f'out must be a dense {x.dtype} tensor of shape {(n, co, oh, ow)} on x\'s device')
        if reduce_x is not None and (reduce_sum is None or reduce_sum.dtype != torch.float32 or tuple(reduce_sum.shape) != (n, co)
                                     or reduce_sum.device != x.device or not reduce_sum.is_contiguous()):
            raise RuntimeError(f'reduce_sum must be a dense float32 tensor of shape {(n, co)} on x\'s device')
    y = out if out is not None else torch.empty([n, co, oh, ow], dtype=x.dtype, device=x.device)
    if y.numel() == 0:
        return y if reduce_x is None else (y, reduce_sum if out is not None else torch.zeros([n, co], device=x.device))
    if reduce_x is not None and (reduce_x.shape != y.shape or reduce_x.dtype != x.dtype or not reduce_x.is_contiguous()):
        raise RuntimeError('reduce_x must be dense and match the output')
    if mask_y is not None and (mask_y.shape != x.shape or mask_y.dtype != x.dtype or not mask_y.is_contiguous()):
        raise RuntimeError('mask_y must be dense and match x')
    if addend is not None and (addend.shape != y.shape or addend.dtype != x.dtype or not addend.is_contiguous()):
        raise RuntimeError('addend must be dense and match the output')
    p = Conv2dParams(x=_lib.ptr(x), w=_lib.ptr(w_packed), y=_lib.ptr(y), in_scale=_lib.ptr(in_scale),
                     out_scale=_lib.ptr(out_scale), bias=_lib.ptr(bias), noise=_lib.ptr(noise),
                     noise_gain=_lib.ptr(noise_gain), noise_stride_n=noise_stride, dtype=_lib.dtype_code(x),
                     n=n, ci=ci, h=h, w_in=w, co=co, oh=oh, ow=ow, in_step=plan.in_step, out_step=plan.out_step,
                     num_phases=len(plan.phases), act=ACT_INDEX[act], alpha=alpha, gain=gain, clamp=clamp,
                     reduce_x=_lib.ptr(reduce_x), reduce_out=None, mask_y=_lib.ptr(mask_y), mask_alpha=mask_alpha,
                     mask_clamp=mask_clamp, addend=_lib.ptr(addend))
    offset = 0
    for i, (oy0, ox0, dy0, dx0, nky, nkx, _, _) in enumerate(plan.phases):
        p.phase[i] = Phase(oy0, ox0, dy0, dx0, nky, nkx, offset)
        offset += nky * nkx * co * ci
    prepared = None
    if x.dtype == torch.float32 and workspace is None and memo.is_constant(w_packed) and not torch.cuda.is_current_stream_capturing():
        # float32 weights that are constants (frozen layers; any layer until its next optimizer step): their largest
        # magnitude and their split rows are kept with them instead of being formed again by every call
        prepared = memo.memo(w_packed, ('float32 prepared', plan.key, co, ci), lambda: _prepare_weights(p, x))
        p.w_prepared = prepared.data_ptr()
    if workspace is not None:
        if workspace.dtype != torch.uint8 or workspace.device != x.device or not workspace.is_contiguous():
            raise RuntimeError('workspace must be a dense uint8 tensor on the device of x')
        p.workspace, p.workspace_bytes = workspace.data_ptr(), workspace.numel()
    else:
        workspace = _attach_workspace(p, _lib.lib().sg_conv2d_workspace_bytes, x)   # (kept alive until the launch is queued)
    partial = None
    if reduce_x is not None and out is not None:
        label = f'{_lib.lib().sg_conv2d_kernel(ctypes.byref(p)).decode()}<{_DTYPE_NAME[x.dtype]}>' if _lib.profiling() else ''
        fn = _lib.lib().sg_conv2d_reduced
        _lib.launch_raw(lambda stream: fn(ctypes.byref(p), _lib.ptr(reduce_sum), stream), x, 'sg_conv2d_reduced', label,
                        _lib.nbytes(x, y, reduce_x))
        return y, reduce_sum
    if reduce_x is not None:
        rows = _lib.lib().sg_conv2d_reduce_rows(ctypes.byref(p))
        if rows < 1:
            raise RuntimeError('sg_conv2d_reduce_rows rejected the geometry')
        partial = torch.zeros([n, rows, co], dtype=torch.float32, device=x.device)
        p.reduce_out = _lib.ptr(partial)
    taps = sum(ph[4] * ph[5] for ph in plan.phases)
    label = f'{_lib.lib().sg_conv2d_kernel(ctypes.byref(p)).decode()}<{_DTYPE_NAME[x.dtype]}>' if _lib.profiling() else ''
    if _DETAIL and label:
        label += f' {plan.kind} k{plan.kh} {ci}->{co} out{oh}x{ow} step{plan.in_step}/{plan.out_step} n{n}'
    _lib.launch('sg_conv2d', p, x, label, _lib.nbytes(x, y, reduce_x), flops=2.0 * n * co * ci * taps * (oh * ow) / len(plan.phases))
    if reduce_x is not None:
        return y, (partial.sum(dim=1) if partial.shape[1] > 1 else partial[:, 0])
    return y


def _prepare_weights(p, like):
    """The buffer `sg_conv2d_prepare_weights` fills for the float32 geometry p (uint8 tensor)."""
    need = int(_lib.lib().sg_conv2d_prepared_weight_bytes(ctypes.byref(p)))
    out = torch.empty([need], dtype=torch.uint8, device=like.device)
    fn = _lib.lib().sg_conv2d_prepare_weights
    _lib.launch_raw(lambda stream: fn(ctypes.byref(p), _lib.ptr(out), need, stream), like, 'sg_conv2d_prepare_weights',
                    'split_rows_kernel', 2 * need)
    return out


def workspace_bytes(x, plan, co):
    """Bytes of scratch memory `conv2d(x, ..., plan, co)` uses (0 for 16-bit tensors)."""
    n, ci, h, w = x.shape
    p = Conv2dParams(dtype=_lib.dtype_code(x), n=n, ci=ci, h=h, w_in=w, co=co, num_phases=len(plan.phases))
    offset = 0
    for i, (oy0, ox0, dy0, dx0, nky, nkx, _, _) in enumerate(plan.phases):
        p.phase[i] = Phase(oy0, ox0, dy0, dx0, nky, nkx, offset)
        offset += nky * nkx * co * ci
    return int(_lib.lib().sg_conv2d_workspace_bytes(ctypes.byref(p)))


def supports_mask(x, plan, co, out_hw=None, reduce=False):
    """True when `conv2d(x, ..., mask_y=...)` has a kernel for this geometry (x: the gradient tensor, co: channels of
    the result)."""
    if not x.is_cuda or x.dim() != 4 or not x.is_contiguous() or x.dtype not in _DTYPE_NAME:
        return False
    n, ci, h, w = x.shape
    oh, ow = plan.out_size(h, w) if out_hw is None else out_hw
    p = Conv2dParams(x=_lib.ptr(x), w=_lib.ptr(x), y=_lib.ptr(x), mask_y=_lib.ptr(x), dtype=_lib.dtype_code(x), n=n, ci=ci, h=h,
                     w_in=w, co=co, oh=oh, ow=ow, in_step=plan.in_step, out_step=plan.out_step, num_phases=len(plan.phases),
                     act=1, gain=1.0, clamp=-1.0, mask_clamp=-1.0)
    if reduce:
        p.reduce_x = _lib.ptr(x)
        p.reduce_out = _lib.ptr(x)
    offset = 0
    for i, (oy0, ox0, dy0, dx0, nky, nkx, _, _) in enumerate(plan.phases):
        p.phase[i] = Phase(oy0, ox0, dy0, dx0, nky, nkx, offset)
        offset += nky * nkx * co * ci
    return bool(_lib.lib().sg_conv2d_supports_mask(ctypes.byref(p)))


def supports_addend(x, plan, co, out_hw=None, epilogue_free=True, reduce=False, mask=False):
    """True when `conv2d(x, ..., addend=...)` has a kernel for this geometry: 1x1 layers without modulation or noise, and
    1x1 layers with three input channels (the data gradient of a to-RGB layer; `reduce`: with the style-gradient
    reduction, `mask`: with an activation mask on the input)."""
    if not x.is_cuda or x.dim() != 4 or not x.is_contiguous() or x.dtype not in _DTYPE_NAME:
        return False
    n, ci, h, w = x.shape
    oh, ow = plan.out_size(h, w) if out_hw is None else out_hw
    p = Conv2dParams(x=_lib.ptr(x), w=_lib.ptr(x), y=_lib.ptr(x), addend=_lib.ptr(x), dtype=_lib.dtype_code(x), n=n, ci=ci, h=h,
                     w_in=w, co=co, oh=oh, ow=ow, in_step=plan.in_step, out_step=plan.out_step, num_phases=len(plan.phases),
                     act=1, gain=1.0, clamp=-1.0, mask_clamp=-1.0, reduce_x=_lib.ptr(x) if reduce else None,
                     mask_y=_lib.ptr(x) if mask else None)
    offset = 0
    for i, (oy0, ox0, dy0, dx0, nky, nkx, _, _) in enumerate(plan.phases):
        p.phase[i] = Phase(oy0, ox0, dy0, dx0, nky, nkx, offset)
        offset += nky * nkx * co * ci
    return bool(_lib.lib().sg_conv2d_supports_addend(ctypes.byref(p)))


def kernel_name(x, plan, co, out_hw=None):
    """Name of the device kernel `conv2d(x, ..., plan, co)` runs for a plain call (no scales, noise, reduction or mask),
    as `sg_conv2d_kernel` reports it."""
    n, ci, h, w = x.shape
    oh, ow = plan.out_size(h, w) if out_hw is None else out_hw
    p = Conv2dParams(x=_lib.ptr(x), w=_lib.ptr(x), y=_lib.ptr(x), dtype=_lib.dtype_code(x), n=n, ci=ci, h=h, w_in=w, co=co,
                     oh=oh, ow=ow, in_step=plan.in_step, out_step=plan.out_step, num_phases=len(plan.phases), act=1, gain=1.0,
                     clamp=-1.0, mask_clamp=-1.0)
    offset = 0
    for i, (oy0, ox0, dy0, dx0, nky, nkx, _, _) in enumerate(plan.phases):
        p.phase[i] = Phase(oy0, ox0, dy0, dx0, nky, nkx, offset)
        offset += nky * nkx * co * ci
    return _lib.lib().sg_conv2d_kernel(ctypes.byref(p)).decode()


class Conv2dWgradParams(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('g', ctypes.c_void_p), ('dw', ctypes.c_void_p),
                ('x_scale', ctypes.c_void_p), ('g_scale', ctypes.c_void_p), ('dtype', ctypes.c_int32),
                ('n', ctypes.c_int32), ('cx', ctypes.c_int32), ('h', ctypes.c_int32), ('w_in', ctypes.c_int32),
                ('cg', ctypes.c_int32), ('gh', ctypes.c_int32), ('gw', ctypes.c_int32),
                ('step', ctypes.c_int32), ('dy0', ctypes.c_int32), ('dx0', ctypes.c_int32),
                ('nky', ctypes.c_int32), ('nkx', ctypes.c_int32),
                ('mask_y', ctypes.c_void_p), ('mask_alpha', ctypes.c_float), ('mask_clamp', ctypes.c_float),
                ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t)]


def _wgrad_params(x, g, kh, kw, stride, padding, x_scale=None, g_scale=None, dw=None, mask_y=None, mask_alpha=1.0, mask_clamp=-1.0):
    n, cx, h, w = x.shape
    _, cg, gh, gw = g.shape
    return Conv2dWgradParams(x=_lib.ptr(x), g=_lib.ptr(g), dw=_lib.ptr(dw), x_scale=_lib.ptr(x_scale), g_scale=_lib.ptr(g_scale),
                             dtype=_lib.dtype_code(x), n=n, cx=cx, h=h, w_in=w, cg=cg, gh=gh, gw=gw, step=stride, dy0=-padding,
                             dx0=-padding, nky=kh, nkx=kw, mask_y=_lib.ptr(mask_y), mask_alpha=mask_alpha, mask_clamp=mask_clamp)


def wgrad_supports_mask(x, g, kh, kw, stride=1, padding=0):
    """True when `conv2d_wgrad(..., mask_y=...)` has a kernel for this geometry."""
    if not (x.is_cuda and g.is_cuda) or x.dim() != 4 or g.dim() != 4 or x.dtype != g.dtype or x.dtype not in _DTYPE_NAME:
        return False
    if not x.is_contiguous() or not g.is_contiguous():
        return False
    p = _wgrad_params(x, g, kh, kw, stride, padding, mask_y=g)
    return bool(_lib.lib().sg_conv2d_wgrad_supports_mask(ctypes.byref(p)))


def conv2d_wgrad(x, g, kh, kw, stride=1, padding=0, x_scale=None, g_scale=None, mask_y=None, mask_alpha=1.0, mask_clamp=-1.0):
    """dw[cg, cx, kh, kw] (float32) = sum_{n,a,b} g[n,cg,a,b] * x[n,cx,a*stride+ky-padding,b*stride+kx-padding],
    with optional per-sample channel scales on both operands and an optional activation mask on g (`mask_y`: as in
    `conv2d`; ask `wgrad_supports_mask` first).  For y = conv2d(x, w, stride, padding) pass
    g = dy; for y = conv_transpose2d(x, w[ci, co]) pass (x=dy, g=x), which yields dw in w's [ci, co] layout."""
    if not (x.is_cuda and g.is_cuda):
        raise RuntimeError('x and g must reside on CUDA device')
    if x.dim() != 4 or g.dim() != 4 or not x.is_contiguous() or not g.is_contiguous():
        raise RuntimeError('x and g must be dense NCHW tensors of rank 4')
    if x.dtype != g.dtype or x.shape[0] != g.shape[0]:
        raise RuntimeError('x and g must agree in dtype and batch size')
    n, cx, h, w = x.shape
    _, cg, gh, gw = g.shape
    for name, t, shape in (('x_scale', x_scale, (n, cx)), ('g_scale', g_scale, (n, cg))):
        if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous()):
            raise RuntimeError(f'{name} must be a dense float32 tensor of shape {shape}')
    dw = torch.zeros([kh, kw, cg, cx], dtype=torch.float32, device=x.device)
    if n > 0:
        if mask_y is not None and (mask_y.shape != g.shape or mask_y.dtype != g.dtype or not mask_y.is_contiguous()):
            raise RuntimeError('mask_y must be dense and match g')
        p = _wgrad_params(x, g, kh, kw, stride, padding, x_scale, g_scale, dw, mask_y, mask_alpha, mask_clamp)
        work = _attach_workspace(p, _lib.lib().sg_conv2d_wgrad_workspace_bytes, x)   # noqa: F841
        label = ''
        if _lib.profiling():
            name = _lib.lib().sg_conv2d_wgrad_kernel(ctypes.byref(p)).decode()
            taps = '' if name in ('wgrad3x3_kernel', 'wgrad_f32_kernel', 'wgrad1x1_few_kernel') else f',{1 if kh * kw <= 1 else 4 if kh * kw <= 4 else 9}'
            label = f'{name}<{_DTYPE_NAME[x.dtype]}{taps}>' if name != 'wgrad_f32_kernel' else name
        if _DETAIL and label:
            label += f' k{kh} {cx}->{cg} g{gh}x{gw} step{stride} n{n}'
        _lib.launch('sg_conv2d_wgrad', p, x, label, _lib.nbytes(x, g), flops=2.0 * n * cg * cx * kh * kw * gh * gw)
    return dw.permute(2, 3, 0, 1)
