"""One-kernel convolution layers: modulation, contraction, demodulation, noise, bias, leaky ReLU,
gain and clamp in a single pass (two passes when the layer up-samples: transposed convolution,
then blur with the epilogue), and a backward of three passes (epilogue backward with its
reductions, data gradient with the style-gradient reduction, weight gradient).

This is the training-path `modulated_conv2d` + `bias_act` of a synthesis layer
(reference: training/networks.py:640-653 and 919-921), a to-RGB layer (:961-962) and a
discriminator `Conv2dLayer` (:748-760) without their intermediate tensors.  When a
higher-order graph is being recorded (path-length and R1 regularisation) the backward re-expresses
the layer with the differentiable primitive ops instead, so gradients of any order stay available; a forward pass
that is known to be differentiated that way runs inside `differentiable_twice()` and skips the one-kernel form.
"""
import contextlib
import weakref

import torch

from ... import conv2d_plugin as _conv
from ... import layer_plugin as _layer
from ... import upfirdn2d_plugin as _fir
from . import bias_act
from . import conv2d_gradfix
from . import planes
from . import upfirdn2d
from .. import memo
from .conv2d_resample import _align_extra
from .upfirdn2d import _get_filter_size


MIN_FUSED_PIXELS_F32 = 1 << 15
ONE_PASS_UP_MAX_CHANNELS = 64


def supported(x, w, up, down, groups=1):
    """Geometries the fused kernels cover; everything else takes the composite path."""
    if not x.is_cuda or x.dtype not in (torch.float32, torch.float16, torch.bfloat16) or w.dtype != x.dtype:
        return False
    kh, kw = int(w.shape[2]), int(w.shape[3])
    if groups != 1 or kh != kw or kh not in (1, 3):
        return False
    # Low-resolution float32 layers have too few pixel tiles to fill the chip; the composite path lets their
    # contraction split the channel sum over workgroups, which a fused epilogue cannot.
    if x.dtype == torch.float32 and x.shape[0] * x.shape[2] * x.shape[3] * (up * up) // (down * down) < MIN_FUSED_PIXELS_F32:
        return False
    return (up, down) in ((1, 1), (1, 2)) or ((up, down) == (2, 1) and kh == 3)


def _flipped(w):
    return memo.memo(w, 'flip', lambda: w.flip([2, 3]))


_FILTER_TAPS = {}   # id(filter tensor) -> (weak reference, version, separable_taps(filter))


def _filter_taps(f):
    hit = _FILTER_TAPS.get(id(f))
    if hit is not None and hit[0]() is f and hit[1] == f._version:
        return hit[2]
    if f.is_cuda and torch.cuda.is_current_stream_capturing():
        return None     # the rank check reads the filter back: not while a graph is being recorded
    value = _layer.separable_taps(f)
    _FILTER_TAPS[id(f)] = (weakref.ref(f, lambda _, key=id(f): _FILTER_TAPS.pop(key, None)), f._version, value)
    return value


def _one_pass_up(x, w, f, k, blur_pad, extra, act, alpha, gain):
    """(composed weights, vertical taps) when the one-pass kernel covers this up-sampling layer, else None."""
    # From 128 input channels the layer is bound by its matrix work, which the one-pass form doubles (18 tap
    # combinations instead of 9): there the two passes are faster (256 -> 128 channels at 128^2, 32 samples: 0.85 ms
    # against 1.10 ms; 128 -> 64 at 256^2: 1.20 against 1.28; 64 -> 32 at 512^2: 2.08 against 1.43).
    if x.shape[1] > ONE_PASS_UP_MAX_CHANNELS:
        return None
    if k != 3 or tuple(blur_pad) != (1, 1 - extra, 1, 1) or not _layer.conv_up2_fir_supported(x, w.shape[0], act, alpha, gain):
        return None
    taps = _filter_taps(f)
    if taps is None:
        return None
    m, fy4 = taps
    # (keyed by the filter's identity and version as well: the composed weights hold its horizontal taps)
    return memo.memo(w, ('up2_fir', f.data_ptr(), f._version), lambda: _layer.compose_up2_weights(w, m)), fy4


_twice = False   # True inside differentiable_twice()


@contextlib.contextmanager
def differentiable_twice():
    """Layers called inside run as the composite of differentiable primitive ops from the start.  For a forward pass
    whose backward will be taken with `create_graph=True` (path-length and R1 regularisation): the one-kernel layer
    would have to be evaluated a second time as that composite when its backward is asked for a graph."""
    global _twice
    old = _twice
    _twice = True
    try:
        yield
    finally:
        _twice = old


def _noise_plane(noise, dtype):
    return None if noise is None else noise.to(torch.float32).contiguous()


def _composite(x, w, styles, dcoefs, noise, noise_strength, bias, f, cfg):
    """The same layer from the differentiable primitive ops."""
    mode, k, stride, padding, out_hw, act, alpha, gain, clamp, blur_pad, extra = cfg
    n = x.shape[0]
    if styles is not None:
        x = planes.scale_planes(x, styles)
    if mode == 'same':
        c = conv2d_gradfix.conv2d(x, w, stride=stride, padding=padding, output_size=out_hw)
    else:
        c = conv2d_gradfix.conv_transpose2d(x, w.transpose(0, 1), stride=2, padding=0, output_padding=[0, extra])
        c = upfirdn2d.upfirdn2d(c, f, padding=list(blur_pad), gain=4)
    nz = None
    if noise is not None:
        nz = (noise * noise_strength).to(x.dtype) if noise_strength is not None else noise.to(x.dtype)
    if dcoefs is not None:
        c = planes.scale_planes(c, dcoefs)
    if nz is not None:
        c = c + nz
    return bias_act.bias_act(c, bias, act=act, alpha=alpha, gain=gain, clamp=clamp if clamp >= 0 else None)


class _FusedConv(torch.autograd.Function):
    """cfg = (mode, k, stride, padding, out_hw, act, alpha, gain, clamp, blur_pad, extra).
    mode 'same': y = epilogue(conv(x * styles, w; stride, padding)[:out_hw]);
    mode 'up2' : y = epilogue(blur(conv_transpose(x * styles, w; stride 2), f, blur_pad, gain 4))."""

    @staticmethod
    def forward(ctx, x, w, styles, dcoefs, noise, noise_strength, bias, f, cfg, addend=None):
        mode, k, stride, padding, out_hw, act, alpha, gain, clamp, blur_pad, extra = cfg
        x = x.contiguous()
        co = w.shape[0]
        ep = dict(noise=noise, noise_gain=noise_strength if noise is not None else None, bias=bias, act=act, alpha=alpha,
                  gain=gain, clamp=clamp)
        if mode == 'same':
            plan = _conv.plan_conv(k, k, stride=stride, padding=padding)
            y = _conv.conv2d(x, plan.pack(w), plan, co, in_scale=styles, out_scale=dcoefs, out_hw=out_hw, addend=addend, **ep)
        else:
            assert addend is None
            plan = _conv.plan_conv_transpose(k, k, stride=2, padding=0)
            th, tw = plan.out_size(x.shape[2], x.shape[3])
            ctx.t_shape = (x.shape[0], co, th, tw + extra)
            fir = _one_pass_up(x, w, f, k, blur_pad, extra, act, alpha, gain)
            if fir is not None:     # transposed convolution, filter and epilogue in one kernel, no intermediate
                y = _layer.conv_up2_fir(x, fir[0], co, fir[1], in_scale=styles, scale=dcoefs, **ep)
            else:
                t = _conv.conv2d(x, plan.pack(w), plan, co, in_scale=styles, out_hw=(th, tw + extra))
                y = _layer.blur_epilogue(t, f, blur_pad, False, 4.0, scale=dcoefs, **ep)
        # the output is only needed where the backward reads it: activation slope, clamp mask, demodulation
        # gradient; a linear, unclamped layer (the residual skips) may then be updated in place by its caller
        keep_y = act != 'linear' or clamp >= 0 or dcoefs is not None
        assert addend is None or not keep_y    # (the sum would hide the layer's own output from its backward)
        ctx.save_for_backward(x, w, styles, dcoefs, noise, noise_strength, bias, f, y if keep_y else None)
        ctx.w_memo = getattr(w, '_sg_memo', None)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        return _backward(ctx, dy, None)


class _FusedConvBranch(torch.autograd.Function):
    """(x, layer(x)) for an input that has a second consumer - the output of a synthesis block feeds its to-RGB layer and
    the next block (training/networks.py:1041-1046).  The gradient of such a tensor is the sum of what its consumers
    return, which autograd forms in a pass of its own (read two tensors, write one).  Here the gradient that arrived for
    the passed-through x is handed to the layer's data-gradient kernel, which adds it while it stores (1x1 layers with
    three output channels; elsewhere the sum is a separate step as before).  Same arguments as `_FusedConv`."""

    @staticmethod
    def forward(ctx, x, w, styles, dcoefs, noise, noise_strength, bias, f, cfg):
        y = _FusedConv.forward(ctx, x, w, styles, dcoefs, noise, noise_strength, bias, f, cfg)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, gx, dy):
        if dy is None:
            return (gx if ctx.needs_input_grad[0] else None,) + (None,) * 8
        return _backward(ctx, dy, gx)[:9]


def _plus(a, b):
    if a is None or b is None:
        return a if b is None else b
    return a + b


def _backward(ctx, dy, gx):
    """Gradients of `_FusedConv` / `_FusedConvBranch`; `gx`: gradient of the passed-through input (or None), added to
    the input's gradient."""
    x, w, styles, dcoefs, noise, noise_strength, bias, f, y = ctx.saved_tensors
    if ctx.w_memo is not None and not hasattr(w, '_sg_memo'):
        w._sg_memo = ctx.w_memo
    cfg = ctx.cfg
    mode, k, stride, padding, out_hw, act, alpha, gain, clamp, blur_pad, extra = cfg
    need = ctx.needs_input_grad   # x, w, styles, dcoefs, noise, noise_strength, bias, f, cfg, addend
    d_addend = dy if len(need) > 9 and need[9] else None

    if torch.is_grad_enabled():   # a higher-order graph is wanted: differentiate the composite instead
        with torch.enable_grad():
            # Differentiate with respect to aliases of the inputs: an input may be an ancestor of another
            # (the demodulation coefficients are computed from the styles and the weight), and the gradient
            # returned for it must hold the direct dependence only - the caller's graph adds the rest.
            alias = lambda t: t.view_as(t) if t is not None else None
            ins = [alias(x), alias(w), alias(styles), alias(dcoefs), None, alias(noise_strength), alias(bias)]
            out = _composite(ins[0], ins[1], ins[2], ins[3], noise, ins[5], ins[6], f, cfg)
            wanted = [i for i in range(7) if need[i] and ins[i] is not None]
            grads = torch.autograd.grad(out, [ins[i] for i in wanted], dy, create_graph=True, allow_unused=True)
        result = [None] * 10
        for i, g in zip(wanted, grads):
            result[i] = g
        if need[0]:
            result[0] = _plus(result[0], gx)
        result[9] = d_addend
        return tuple(result)

    ep = dict(scale=dcoefs, noise=noise, noise_gain=noise_strength if noise is not None else None, bias=bias, act=act,
              alpha=alpha, gain=gain, clamp=clamp)
    want = dict(want_dbias=need[6] and bias is not None, want_dscale=need[3] and dcoefs is not None,
                want_dnoise_gain=need[5] and noise is not None and noise_strength is not None)
    dx = dw = dstyles = None
    ci = x.shape[1]
    want_ds = need[2] and styles is not None
    want_dx = need[0] or want_ds
    dy = dy.contiguous()
    # The activation mask applied to dy while the kernels stage it: no dz tensor.  Taken for 1x1 layers when a
    # masking kernel exists for every gradient that is wanted (data gradient, weight gradient) at this geometry.
    # (3x3 layers: the masking variants of the band and row kernels read dy and y and mask in registers beside
    # their matrix instructions; writing dz once and reading it twice measured faster - 6.0 against 8.1 ms for
    # the 32-channel layer at 1024 x 1024, 32 samples.)
    if mode == 'same' and k == 1 and y is not None and (want_dx or need[1]):
        plan = _conv.plan_conv_transpose(k, k, stride=stride, padding=padding)
        ok = not want_dx or (tuple(x.shape[2:]) == plan.out_size(*dy.shape[2:]) and _conv.supports_mask(dy, plan, ci, reduce=want_ds))
        ok = ok and (not need[1] or _conv.wgrad_supports_mask(x, dy, k, k, stride=stride, padding=padding))
        if ok:
            mask = dict(mask_y=y, mask_alpha=alpha if act == 'lrelu' else 1.0, mask_clamp=clamp)
            ddcoefs = dng = dbias = None
            if any(want.values()):
                _, dbias, ddcoefs, dng = _layer.epilogue_backward(dy, y, want_dz=False, **want, **ep)
            if want_dx:
                add = None
                if gx is not None and need[0] and _conv.supports_addend(dy, plan, ci, reduce=want_ds, mask=True):
                    add, gx = gx.contiguous(), None      # (the kernel adds it while it stores)
                dx = _conv.conv2d(dy, plan.pack(w, transposed_layout=True), plan, ci, in_scale=dcoefs, out_scale=styles, gain=gain,
                                  reduce_x=x if want_ds else None, addend=add, **mask)
                if want_ds:
                    dx, dstyles = dx
                    dstyles = dstyles * gain      # the reduction is taken before the output scale and gain
            if need[1]:
                dw = (_conv.conv2d_wgrad(x, dy, k, k, stride=stride, padding=padding, x_scale=styles, g_scale=dcoefs, **mask)
                      * gain).to(w.dtype)
            return (_plus(dx, gx) if need[0] else None, dw, dstyles, ddcoefs, None,
                    dng.reshape(noise_strength.shape) if dng is not None else None,
                    dbias.to(bias.dtype) if dbias is not None else None, None, None, d_addend)
    dz, dbias, ddcoefs, dng = _layer.epilogue_backward(dy, y if y is not None else dy, **want, **ep)
    if mode == 'same':
        if want_dx:
            plan = _conv.plan_conv_transpose(k, k, stride=stride, padding=padding)
            dx = _conv.conv2d(dz, plan.pack(w, transposed_layout=True), plan, ci, out_scale=styles,
                              out_hw=tuple(x.shape[2:]), reduce_x=x if want_ds else None)
        if need[1]:
            dw = _conv.conv2d_wgrad(x, dz, k, k, stride=stride, padding=padding, x_scale=styles).to(w.dtype)
    else:
        fw, fh = _get_filter_size(f)
        _, _, th, tw = ctx.t_shape
        oh, ow = dz.shape[2:]
        px0, px1, py0, py1 = blur_pad
        dt = _fir.upfirdn2d(dz, f, 1, 1, 1, 1, fw - px0 - 1, tw - ow + px0, fh - py0 - 1, th - oh + py0, True, 4.0)
        if want_dx:
            plan = _conv.plan_conv(k, k, stride=2, padding=0)
            packed = memo.memo(w, ('pack^T', plan.key), lambda: plan.pack(w.transpose(0, 1)))
            dx = _conv.conv2d(dt, packed, plan, ci, out_scale=styles, out_hw=tuple(x.shape[2:]),
                              reduce_x=x if want_ds else None)
        if need[1]:   # weight of the transposed conv is w^T [ci, co]; exchange the operands' roles
            dw = _conv.conv2d_wgrad(dt, x, k, k, stride=2, padding=0, g_scale=styles).transpose(0, 1).to(w.dtype)
    if want_ds and dx is not None:
        dx, dstyles = dx
    dx = _plus(dx, gx) if need[0] else None
    return (dx, dw, dstyles, ddcoefs, None, dng.reshape(noise_strength.shape) if dng is not None else None,
            dbias.to(bias.dtype) if dbias is not None else None, None, None, d_addend)


def _down_padding(f, k, padding, down=2):
    fw, fh = _get_filter_size(f)
    pad = int(padding)
    return [pad + (fw - down + 1) // 2, pad + (fw - down) // 2, pad + (fh - down + 1) // 2, pad + (fh - down) // 2]


def branch_down(x, f, padding=0):
    """(x, x filtered and down-sampled by 2) for the input of a down-sampling 1x1 layer that is consumed a second time
    (a residual block's `skip` beside its `conv0`): pass the first result on to the other consumer and the second to
    `conv_layer(..., down=2, prefiltered=True)`.  The two gradients of x then meet inside the filter's backward kernel
    instead of in a sum of their own."""
    return upfirdn2d.branch(x, f, down=2, padding=_down_padding(f, 1, padding))


def conv_layer(x, w, f=None, up=1, down=1, padding=0, flip_weight=True, styles=None, dcoefs=None, noise=None,
               noise_strength=None, bias=None, act='linear', alpha=None, gain=None, clamp=None, prefiltered=False, addend=None,
               also_input=False):
    """clamp(act(conv2d_resample(x * styles, w, f, up, down, padding) * dcoefs + noise * noise_strength + bias) * gain).

    x [N, I, H, W]; w [O, I, k, k] in x's dtype; styles [N, I] / dcoefs [N, O] float32 or None; noise a
    float32 plane [H', W'] or [N, 1, H', W'] (unit noise) with its scalar strength; bias [O] in x's dtype.
    Call `supported()` first.  `addend` (the output's shape and dtype) is added to the result: by the layer's own
    kernel while it stores where one does that (16-bit 1x1 layers without modulation, activation or clamp - the skip
    of a residual block), else in place afterwards.  `also_input` (same-size layers without addend): returns (x, y) -
    pass that x on to the other consumer of the input, and the two gradients of x meet inside this layer's backward
    (`_FusedConvBranch`).
    """
    spec = bias_act.activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    k = int(w.shape[2])
    pad = int(padding)
    styles = styles.to(torch.float32).contiguous() if styles is not None else None
    dcoefs = dcoefs.to(torch.float32).contiguous() if dcoefs is not None else None
    noise = _noise_plane(noise, x.dtype)
    if noise_strength is not None:
        noise_strength = noise_strength.to(torch.float32)
    if bias is not None:
        bias = bias.contiguous()
    fw, fh = _get_filter_size(f)

    if up == 1 and down == 1:
        if not flip_weight:
            w = _flipped(w)
        out_hw = (x.shape[2] + 2 * pad - k + 1, x.shape[3] + 2 * pad - k + 1)
        cfg = ('same', k, 1, pad, out_hw, act, alpha, gain, clamp, None, 0)
    elif down == 2 and up == 1:
        # resampling filter first (differentiable op of its own), then a strided contraction with the epilogue
        px0 = pad + (fw - down + 1) // 2
        px1 = pad + (fw - down) // 2
        py0 = pad + (fh - down + 1) // 2
        py1 = pad + (fh - down) // 2
        if not flip_weight:
            w = _flipped(w)
        if k == 1:
            if not prefiltered:     # (else x is the second result of branch_down)
                x = upfirdn2d.upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1])
            cfg = ('same', 1, 1, 0, tuple(x.shape[2:]), act, alpha, gain, clamp, None, 0)
        else:
            out_hw = ((x.shape[2] + py0 + py1 - fh + 1 - k) // down + 1, (x.shape[3] + px0 + px1 - fw + 1 - k) // down + 1)
            extra = _align_extra(x.shape[3] + px0 + px1 - fw + 1)
            x = upfirdn2d.upfirdn2d(x, f, padding=[px0, px1 + extra, py0, py1])
            cfg = ('same', k, down, 0, out_hw, act, alpha, gain, clamp, None, 0)
    else:   # up == 2, k == 3: transposed convolution (true convolution when flip_weight is False), then blur
        if flip_weight:
            w = _flipped(w)
        px0 = pad + (fw + up - 1) // 2 - (k - 1)
        px1 = pad + (fw - up) // 2 - (k - up)
        py0 = pad + (fh + up - 1) // 2 - (k - 1)
        py1 = pad + (fh - up) // 2 - (k - up)
        assert max(min(-px0, -px1), 0) == 0 and max(min(-py0, -py1), 0) == 0
        extra = _align_extra((x.shape[3] - 1) * up + k)
        cfg = ('up2', k, 2, 0, None, act, alpha, gain, clamp, (px0, px1 - extra, py0, py1), extra)
    if also_input:
        assert addend is None and up == 1 and down == 1
        if _twice and torch.is_grad_enabled():
            return x, _composite(x, w, styles, dcoefs, noise, noise_strength, bias, f, cfg)
        return _FusedConvBranch.apply(x, w, styles, dcoefs, noise, noise_strength, bias, f, cfg)
    if _twice and torch.is_grad_enabled():
        y = _composite(x, w, styles, dcoefs, noise, noise_strength, bias, f, cfg)
        return y if addend is None else y.add_(addend)
    if addend is not None:
        fused = (cfg[0] == 'same' and k == 1 and styles is None and dcoefs is None and noise is None and act == 'linear'
                 and clamp < 0 and addend.is_contiguous()
                 and _conv.supports_addend(x.contiguous(), _conv.plan_conv(1, 1), int(w.shape[0]), out_hw=cfg[4]))
        if not fused:
            return _FusedConv.apply(x, w, styles, dcoefs, noise, noise_strength, bias, f, cfg).add_(addend)
    return _FusedConv.apply(x, w, styles, dcoefs, noise, noise_strength, bias, f, cfg, addend)
