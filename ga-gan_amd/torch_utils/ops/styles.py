"""Styles and demodulation coefficients of a modulated layer as one operator:

    styles = (w @ (a * a_gain).T + b * b_gain) * gain            w [N, K] a row of the latent codes, a [C, K], b [C]
    dcoefs = rsqrt(styles**2 @ wsq_t + eps)                      wsq_t [C, O] = sum over taps of W**2, transposed

`styles = self.affine(w) * weight_gain` of the reference's w_to_s (DissimilarDomains/training/networks.py:502) and the
`dcoefs` of its modulated_conv2d (:631-636).  The forward pass is two launches (`sg_styles`) where the framework's
operators take five, the backward two (`sg_styles_backward`) instead of about fifteen; a backward that is asked for a
graph of its own (path-length regularisation) differentiates the same expressions written with the framework's
operators."""
import ctypes

import torch

from ... import _lib

i32, i64, f32, vp = _lib.i32, _lib.i64, _lib.f32, _lib.vp


class StylesParams(ctypes.Structure):
    _fields_ = [('w', vp), ('a', vp), ('b', vp), ('wsq_t', vp),
                ('a_gain', f32), ('b_gain', f32), ('gain', f32), ('eps', f32),
                ('n', i32), ('k', i32), ('c', i32), ('o', i32), ('w_stride', i64),
                ('styles', vp), ('dcoefs', vp)]


class StylesBwdParams(ctypes.Structure):
    _fields_ = [('f', StylesParams), ('dstyles', vp), ('ddcoefs', vp), ('t', vp), ('dw', vp), ('da', vp), ('db', vp), ('dwsq_t', vp)]


EPS = 1e-8


def supported(w, a, b, wsq_t):
    """float32 CUDA tensors, rows of w at unit stride, K and O multiples of 4."""
    tensors = [t for t in (w, a, b, wsq_t) if t is not None]
    return (all(t.is_cuda and t.dtype == torch.float32 for t in tensors) and w.dim() == 2 and w.stride(1) == 1
            and w.shape[1] % 4 == 0 and w.shape[0] > 0 and a.dim() == 2 and (wsq_t is None or wsq_t.shape[1] % 4 == 0))


def reference(w, a, b, wsq_t, a_gain, b_gain, gain):
    """The same from the framework's operators (differentiable to any order)."""
    styles = torch.addmm((b * b_gain).unsqueeze(0), w, (a * a_gain).t()) if b is not None else w.matmul((a * a_gain).t())
    if gain != 1:
        styles = styles * gain
    dcoefs = None
    if wsq_t is not None:
        dcoefs = (styles.square().matmul(wsq_t) + EPS).rsqrt()
    return styles, dcoefs


def _forward_params(w, a, b, wsq_t, gains, styles, dcoefs):
    n, k = w.shape
    return StylesParams(_lib.ptr(w), _lib.ptr(a), _lib.ptr(b), _lib.ptr(wsq_t), gains[0], gains[1], gains[2], EPS,
                        n, k, a.shape[0], wsq_t.shape[1] if wsq_t is not None else 0, w.stride(0),
                        _lib.ptr(styles), _lib.ptr(dcoefs))


class _Styles(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, a, b, wsq_t, a_gain, b_gain, gain):
        a = a.contiguous()
        b = b.contiguous() if b is not None else None
        wsq_t = wsq_t.contiguous() if wsq_t is not None else None
        n, c = w.shape[0], a.shape[0]
        styles = torch.empty([n, c], dtype=torch.float32, device=w.device)
        dcoefs = torch.empty([n, wsq_t.shape[1]], dtype=torch.float32, device=w.device) if wsq_t is not None else None
        p = _forward_params(w, a, b, wsq_t, (a_gain, b_gain, gain), styles, dcoefs)
        fn = _lib.lib().sg_styles
        _lib.launch_raw(lambda stream: fn(ctypes.byref(p), stream), w, 'sg_styles', 'styles_kernel', _lib.nbytes(w, a, wsq_t))
        ctx.save_for_backward(w, a, b, wsq_t, styles, dcoefs)
        ctx.set_materialize_grads(False)
        ctx.gains = (a_gain, b_gain, gain)
        return styles, dcoefs

    @staticmethod
    def backward(ctx, dstyles, ddcoefs):
        w, a, b, wsq_t, styles, dcoefs = ctx.saved_tensors
        need = ctx.needs_input_grad
        if dstyles is None and ddcoefs is None:
            return (None,) * 7
        if torch.is_grad_enabled():   # a higher-order graph is wanted: differentiate the operators' version
            with torch.enable_grad():
                ins = [t.view_as(t) if t is not None else None for t in (w, a, b, wsq_t)]
                out = reference(*ins, *ctx.gains)
                pairs = [(o, g) for o, g in zip(out, (dstyles, ddcoefs)) if o is not None and g is not None]
                wanted = [i for i in range(4) if need[i] and ins[i] is not None]
                grads = torch.autograd.grad([o for o, _ in pairs], [ins[i] for i in wanted], [g for _, g in pairs],
                                            create_graph=True, allow_unused=True)
            result = [None] * 7
            for i, g in zip(wanted, grads):
                result[i] = g
            return tuple(result)
        dstyles = dstyles.contiguous() if dstyles is not None else None
        ddcoefs = ddcoefs.contiguous() if ddcoefs is not None and dcoefs is not None else None
        new = lambda like: torch.empty(like.shape, dtype=torch.float32, device=w.device)
        t = new(styles)
        dw = torch.empty(list(w.shape), dtype=torch.float32, device=w.device) if need[0] else None
        da = new(a) if need[1] else None
        db = new(b) if need[2] and b is not None else None
        dwsq = None
        if need[3] and wsq_t is not None:
            dwsq = new(wsq_t) if ddcoefs is not None else torch.zeros_like(wsq_t)
        f = _forward_params(w, a, b, wsq_t, ctx.gains, styles, dcoefs)
        p = StylesBwdParams(f, _lib.ptr(dstyles), _lib.ptr(ddcoefs), _lib.ptr(t), _lib.ptr(dw), _lib.ptr(da), _lib.ptr(db),
                            _lib.ptr(dwsq) if ddcoefs is not None else vp(None))
        fn = _lib.lib().sg_styles_backward
        _lib.launch_raw(lambda stream: fn(ctypes.byref(p), stream), w, 'sg_styles_backward', 'styles_grads_kernel',
                        _lib.nbytes(w, a, wsq_t))
        return dw, da, db, dwsq, None, None, None


def styles(w, a, b, wsq_t=None, a_gain=1.0, b_gain=1.0, gain=1.0):
    """(styles [N, C], dcoefs [N, O] or None)."""
    if not supported(w, a, b, wsq_t):
        return reference(w, a, b, wsq_t, a_gain, b_gain, gain)
    return _Styles.apply(w, a, b, wsq_t, float(a_gain), float(b_gain), float(gain))
