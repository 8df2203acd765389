"""Fused bias + activation + gain + clamp on the GPU, with first and second order
gradients.  Same public surface as the reference's torch_utils/ops/bias_act.py
(`activation_funcs`, `bias_act`), executed by the gfx950 kernel behind
`ga_gan_amd.bias_act_plugin`."""
import math
from types import SimpleNamespace

import torch

from ... import bias_act_plugin as _plugin
from .. import misc
from . import conv2d_gradfix


def _spec(func, def_alpha, def_gain, cuda_idx, ref, has_2nd_grad):
    return SimpleNamespace(func=func, def_alpha=def_alpha, def_gain=def_gain, cuda_idx=cuda_idx, ref=ref,
                           has_2nd_grad=has_2nd_grad)


# Activation table (reference: bias_act.py:23-60).  `ref` names what the backward needs
# (saved output y, or input x), `cuda_idx` is the kernel's activation index.
activation_funcs = {
    'linear':   _spec(lambda x, **_: x, 0, 1, 1, '', False),
    'relu':     _spec(lambda x, **_: torch.nn.functional.relu(x), 0, math.sqrt(2), 2, 'y', False),
    'lrelu':    _spec(lambda x, alpha, **_: torch.nn.functional.leaky_relu(x, alpha), 0.2, math.sqrt(2), 3, 'y', False),
    'tanh':     _spec(lambda x, **_: torch.tanh(x), 0, 1, 4, 'y', True),
    'sigmoid':  _spec(lambda x, **_: torch.sigmoid(x), 0, 1, 5, 'y', True),
    'elu':      _spec(lambda x, **_: torch.nn.functional.elu(x), 0, 1, 6, 'y', True),
    'selu':     _spec(lambda x, **_: torch.nn.functional.selu(x), 0, 1, 7, 'y', True),
    'softplus': _spec(lambda x, **_: torch.nn.functional.softplus(x), 0, 1, 8, 'y', True),
    'swish':    _spec(lambda x, **_: torch.sigmoid(x) * x, 0, math.sqrt(2), 9, 'x', True),
}

_null = torch.empty([0])


def _null_like(x):
    return torch.empty([0], dtype=x.dtype, device=x.device)


def _layout(x):
    return torch.channels_last if x.ndim > 2 and x.stride(1) == 1 and x.shape[1] > 1 else torch.contiguous_format


def _sum_except(t, dim):
    return t.sum([i for i in range(t.ndim) if i != dim])


class _BiasAct(torch.autograd.Function):
    """cfg = (dim, act, alpha, gain, clamp) with clamp < 0 meaning off."""

    @staticmethod
    def forward(ctx, x, b, cfg):
        dim, act, alpha, gain, clamp = cfg
        spec = activation_funcs[act]
        ctx.cfg = cfg
        ctx.memory_format = _layout(x)
        x = x.contiguous(memory_format=ctx.memory_format)
        null = _null_like(x)
        bb = b.contiguous() if b is not None else null
        y = x
        if act != 'linear' or gain != 1 or clamp >= 0 or b is not None:
            y = _plugin.bias_act(x, bb, null, null, null, 0, dim, spec.cuda_idx, alpha, gain, clamp)
        needs_x = 'x' in spec.ref or spec.has_2nd_grad
        # The clamp mask of the backward is taken from the saved output, so y is kept whenever a
        # clamp is active - also for 'linear', whose derivative the reference's CPU path masks too
        # (its CUDA plugin does not: bias_act.py:193-196 saves no y for 'linear').
        needs_y = 'y' in spec.ref or (clamp >= 0 and 'x' not in spec.ref)
        ctx.save_for_backward(x if needs_x else null, bb if needs_x else null, y if needs_y else null)
        return y

    @staticmethod
    def backward(ctx, dy):
        dim, act, alpha, gain, clamp = ctx.cfg
        dy = dy.contiguous(memory_format=ctx.memory_format)
        x, b, y = ctx.saved_tensors
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy
            if act != 'linear' or gain != 1 or clamp >= 0:
                dx = _BiasActGrad.apply(dy, x, b, y, ctx.cfg)
        # (inside conv2d_gradfix.no_weight_gradients() - the first-order passes of the two regularisers, which ask for the
        # gradient of an activation only - the bias gradient is a reduction over the whole tensor that nobody reads)
        if ctx.needs_input_grad[1] and not conv2d_gradfix.parameter_gradients_disabled:
            db = _sum_except(dx, dim)
        return dx, db, None


class _BiasActGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, cfg):
        dim, act, alpha, gain, clamp = cfg
        spec = activation_funcs[act]
        ctx.cfg = cfg
        ctx.memory_format = _layout(dy)
        null = _null_like(dy)
        dx = _plugin.bias_act(dy, b, x, y, null, 1, dim, spec.cuda_idx, alpha, gain, clamp)
        ctx.save_for_backward(dy if spec.has_2nd_grad else null, x, b, y)
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dim, act, alpha, gain, clamp = ctx.cfg
        spec = activation_funcs[act]
        d_dx = d_dx.contiguous(memory_format=ctx.memory_format)
        dy, x, b, y = ctx.saved_tensors
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, ctx.cfg)
        if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _plugin.bias_act(d_dx, b, x, y, dy, 2, dim, spec.cuda_idx, alpha, gain, clamp)
        if spec.has_2nd_grad and ctx.needs_input_grad[2]:
            d_b = _sum_except(d_x, dim)
        return d_dy, d_x, d_b, None, None


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(act(x + b) * gain); arguments as the reference's `bias_act`
    (torch_utils/ops/bias_act.py:88-122).  There is no `impl='ref'` in the product path."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl != 'cuda':
        raise NotImplementedError("impl='ref' is not part of this build (the CPU restatement is the test oracle)")
    misc.require_gpu(x, 'bias_act')
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clamp = float(clamp if clamp is not None else -1)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
    return _BiasAct.apply(x, b, (dim, act, alpha, gain, clamp))
