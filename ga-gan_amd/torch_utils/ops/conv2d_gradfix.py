"""2-D convolution entry points of the op API (reference: torch_utils/ops/conv2d_gradfix.py),
running on the matrix-core kernels `sg_conv2d` / `sg_conv2d_wgrad`.

`conv2d` and `conv_transpose2d` are three mutually recursive autograd functions - forward,
data gradient (the transposed operation) and weight gradient - so gradients of any order
stay on the same kernels (the structure of the reference's custom op,
conv2d_gradfix.py:96-214; there it is only active on torch 1.7-1.9).
`no_weight_gradients()` skips the weight-gradient kernels inside the regularisation passes
(reference: training/loss.py:98,142).

Shapes the kernels do not cover (groups > 1, dilation, asymmetric stride / padding, strides
above 2, kernels above 4x4 taps, float64) go to the library convolution of PyTorch-ROCm.
"""
import contextlib

import torch

from ... import conv2d_plugin as _plugin

enabled = True                      # kept for callers that set it (training_loop.py:209 of the reference)
weight_gradients_disabled = False   # True inside no_weight_gradients()


parameter_gradients_disabled = False   # True inside no_weight_gradients(all_parameters=True)


@contextlib.contextmanager
def no_weight_gradients(all_parameters=False):
    """Inside, convolutions return no weight gradient (reference: conv2d_gradfix.py:26-32).  With
    `all_parameters=True` the bias gradient of `bias_act` and the addend gradient of `fma` (whole-tensor reductions)
    are skipped as well: for callers that differentiate with respect to inputs only, as the loss's two
    regularisers do.  Without it those two ops behave as the reference's do inside this context."""
    global weight_gradients_disabled, parameter_gradients_disabled
    old = weight_gradients_disabled, parameter_gradients_disabled
    weight_gradients_disabled = True
    parameter_gradients_disabled = parameter_gradients_disabled or all_parameters
    try:
        yield
    finally:
        weight_gradients_disabled, parameter_gradients_disabled = old


def _pair(v):
    if isinstance(v, (tuple, list)):
        assert len(v) == 2
        return int(v[0]), int(v[1])
    return int(v), int(v)


def _native_cfg(input, weight, bias, stride, padding, output_padding, dilation, groups, transpose):
    """(stride, padding, output_padding, kh, kw) when the native kernels cover the call, else None."""
    if not enabled or bias is not None or groups != 1 or not input.is_cuda:
        return None
    if input.dtype not in (torch.float32, torch.float16, torch.bfloat16) or weight.dtype != input.dtype:
        return None
    if input.ndim != 4 or weight.ndim != 4:
        return None
    (sy, sx), (py, px), (oy, ox), (dy, dx) = _pair(stride), _pair(padding), _pair(output_padding), _pair(dilation)
    kh, kw = int(weight.shape[2]), int(weight.shape[3])
    if sy != sx or py != px or oy < 0 or ox < 0 or (dy, dx) != (1, 1) or sy not in (1, 2) or kh * kw > 9 or py < 0:
        return None
    if transpose and py > min(kh, kw) - 1:
        return None
    return sy, py, (oy, ox), kh, kw


class _Conv2d(torch.autograd.Function):
    """cfg = (transpose, stride, padding, out_hw, kh, kw).  transpose=False: weight is [co, ci, kh, kw];
    transpose=True: weight is [ci, co, kh, kw] (torch's conv_transpose2d layout).  out_hw is the output
    size: for a transposed convolution it fixes the output padding, for a plain one it may be smaller
    than the natural size (trailing input rows / columns are then never read)."""

    @staticmethod
    def forward(ctx, input, weight, cfg):
        transpose, stride, padding, out_hw, kh, kw = cfg
        x = input.contiguous()
        if not transpose:
            plan = _plugin.plan_conv(kh, kw, stride=stride, padding=padding)
            y = _plugin.conv2d(x, plan.pack(weight), plan, weight.shape[0], out_hw=out_hw)
        else:
            plan = _plugin.plan_conv_transpose(kh, kw, stride=stride, padding=padding)
            y = _plugin.conv2d(x, plan.pack(weight, transposed_layout=True), plan, weight.shape[1], out_hw=out_hw)
        ctx.save_for_backward(input, weight)
        ctx.w_memo = getattr(weight, '_sg_memo', None)    # packings memoised on a constant weight (torch_utils/memo.py)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        if ctx.w_memo is not None and not hasattr(weight, '_sg_memo'):
            weight._sg_memo = ctx.w_memo
        transpose, stride, padding, out_hw, kh, kw = ctx.cfg
        grad_input = grad_weight = None
        if ctx.needs_input_grad[0]:
            if not transpose:   # rows / columns the strided window never reached get zero gradient
                grad_input = _Conv2d.apply(grad_output, weight, (True, stride, padding, tuple(input.shape[2:]), kh, kw))
            else:
                grad_input = _Conv2d.apply(grad_output, weight, (False, stride, padding, tuple(input.shape[2:]), kh, kw))
            assert grad_input.shape == input.shape
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            grad_weight = _Conv2dGradWeight.apply(grad_output, input, ctx.cfg, weight.shape)
        return grad_input, grad_weight, None


class _Conv2dGradWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_output, input, cfg, weight_shape):
        transpose, stride, padding, out_hw, kh, kw = cfg
        go, x = grad_output.contiguous(), input.contiguous()
        if not transpose:   # dw[co, ci] = sum dy[co] * x[ci, shifted]
            dw = _plugin.conv2d_wgrad(x, go, kh, kw, stride=stride, padding=padding)
        else:               # dw[ci, co] = sum x[ci] * dy[co, shifted]
            dw = _plugin.conv2d_wgrad(go, x, kh, kw, stride=stride, padding=padding)
        assert dw.shape == weight_shape
        ctx.save_for_backward(grad_output, input)
        ctx.cfg = cfg
        return dw.to(input.dtype)

    @staticmethod
    def backward(ctx, grad2_grad_weight):
        grad_output, input = ctx.saved_tensors
        transpose, stride, padding, out_hw, kh, kw = ctx.cfg
        grad2_grad_output = grad2_input = None
        if ctx.needs_input_grad[0]:
            grad2_grad_output = _Conv2d.apply(input, grad2_grad_weight, ctx.cfg)
            assert grad2_grad_output.shape == grad_output.shape
        if ctx.needs_input_grad[1]:
            if not transpose:
                grad2_input = _Conv2d.apply(grad_output, grad2_grad_weight,
                                            (True, stride, padding, tuple(input.shape[2:]), kh, kw))
            else:
                grad2_input = _Conv2d.apply(grad_output, grad2_grad_weight,
                                            (False, stride, padding, tuple(input.shape[2:]), kh, kw))
            assert grad2_input.shape == input.shape
        return grad2_grad_output, grad2_input, None, None


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, output_size=None):
    """`output_size=(oh, ow)` (an extension of the reference's signature) computes only the leading part of
    the output; the input rows / columns beyond its footprint are ignored and receive zero gradient."""
    cfg = _native_cfg(input, weight, bias, stride, padding, 0, dilation, groups, transpose=False)
    if cfg is not None:
        s, p, _, kh, kw = cfg
        natural = ((input.shape[2] + 2 * p - kh) // s + 1, (input.shape[3] + 2 * p - kw) // s + 1)
        out_hw = natural if output_size is None else tuple(int(v) for v in output_size)
        if 1 <= out_hw[0] <= natural[0] and 1 <= out_hw[1] <= natural[1]:
            return _Conv2d.apply(input, weight, (False, s, p, out_hw, kh, kw))
    assert output_size is None, 'output_size needs the native convolution kernel'
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                      dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    """The native kernel accepts any output_padding >= 0 (the extra rows / columns are zero)."""
    cfg = _native_cfg(input, weight, bias, stride, padding, output_padding, dilation, groups, transpose=True)
    if cfg is not None:
        s, p, op, kh, kw = cfg
        out_hw = ((input.shape[2] - 1) * s + kh - 2 * p + op[0], (input.shape[3] - 1) * s + kw - 2 * p + op[1])
        if out_hw[0] >= 1 and out_hw[1] >= 1:
            return _Conv2d.apply(input, weight, (True, s, p, out_hw, kh, kw))
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)
