"""Oracle (test infrastructure): bias + activation + gain + clamp, fused
multiply-add and minibatch standard deviation on CPU.

Restates:
  * bias_act / activation table   DissimilarDomains/torch_utils/ops/bias_act.py:23-60, 128-157
    and the kernel's gradient conventions torch_utils/ops/bias_act.cu:38-146
  * fma                           torch_utils/ops/fma.py:15-45
  * MinibatchStdLayer.forward     training/networks.py:1284-1301
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# name -> (kernel index, default alpha, default gain); bias_act.py:23-60
ACTIVATIONS = {
    'linear':   (1, 0.0, 1.0),
    'relu':     (2, 0.0, math.sqrt(2)),
    'lrelu':    (3, 0.2, math.sqrt(2)),
    'tanh':     (4, 0.0, 1.0),
    'sigmoid':  (5, 0.0, 1.0),
    'elu':      (6, 0.0, 1.0),
    'selu':     (7, 0.0, 1.0),
    'softplus': (8, 0.0, 1.0),
    'swish':    (9, 0.0, math.sqrt(2)),
}


def _activate(x, act, alpha):
    if act == 'linear':
        return x
    if act == 'relu':
        return F.relu(x)
    if act == 'lrelu':
        return F.leaky_relu(x, alpha)
    if act == 'tanh':
        return torch.tanh(x)
    if act == 'sigmoid':
        return torch.sigmoid(x)
    if act == 'elu':
        return F.elu(x)
    if act == 'selu':
        return F.selu(x)
    if act == 'softplus':
        return F.softplus(x)
    if act == 'swish':
        return torch.sigmoid(x) * x
    raise KeyError(act)


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """clamp(act(x + b) * gain, +-clamp); defaults per activation."""
    _, def_alpha, def_gain = ACTIVATIONS[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = _activate(x, act, alpha)
    if gain != 1:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def bias_act_numpy(x, b, xref, yref, dy, grad, step_b, act_idx, alpha, gain, clamp):
    """float64 element-wise statement of the native entry point's three modes
    (grad = 0 forward, 1 first-order, 2 second-order; bias_act.cu:38-146).
    Flat arrays; absent operands are None.  Used to pin the `grad` modes of the
    HIP kernel, which the differentiable `bias_act` above does not expose."""
    x = np.asarray(x, dtype=np.float64).ravel().copy()
    n = x.size
    bb = np.zeros(n) if b is None else np.asarray(b, np.float64)[(np.arange(n) // step_b) % len(b)]
    xr = np.zeros(n) if xref is None else np.asarray(xref, np.float64).ravel().copy()
    yr = np.zeros(n) if yref is None else np.asarray(yref, np.float64).ravel().copy()
    dyv = np.ones(n) if dy is None else np.asarray(dy, np.float64).ravel()
    yy = yr / gain if gain != 0 else np.zeros(n)
    if grad == 0:
        x = x + bb
    else:
        xr = xr + bb
    G, A = grad, act_idx
    y = np.zeros(n)
    sel_scale, sel_alpha = 1.0507009873554804934193349852946, 1.6732632423543772848170429916717
    with np.errstate(over='ignore', invalid='ignore'):
        if A == 1:
            y = x if G in (0, 1) else y
        elif A == 2:
            y = np.where(x > 0, x, 0) if G == 0 else np.where(yy > 0, x, 0) if G == 1 else y
        elif A == 3:
            y = np.where(x > 0, x, x * alpha) if G == 0 else np.where(yy > 0, x, x * alpha) if G == 1 else y
        elif A == 4:
            y = np.tanh(x) if G == 0 else x * (1 - yy * yy) if G == 1 else x * (1 - yy * yy) * (-2 * yy)
        elif A == 5:
            y = 1 / (np.exp(-x) + 1) if G == 0 else x * yy * (1 - yy) if G == 1 else x * yy * (1 - yy) * (1 - 2 * yy)
        elif A == 6:
            y = (np.where(x >= 0, x, np.exp(x) - 1) if G == 0 else
                 np.where(yy >= 0, x, x * (yy + 1)) if G == 1 else np.where(yy >= 0, 0, x * (yy + 1)))
        elif A == 7:
            y = (np.where(x >= 0, sel_scale * x, sel_scale * sel_alpha * (np.exp(x) - 1)) if G == 0 else
                 np.where(yy >= 0, x * sel_scale, x * (yy + sel_scale * sel_alpha)) if G == 1 else
                 np.where(yy >= 0, 0, x * (yy + sel_scale * sel_alpha)))
        elif A == 8:
            if G == 0:
                y = np.where(x > 80, x, np.log(np.exp(x) + 1))
            else:
                c = np.exp(-yy)
                y = x * (1 - c) if G == 1 else x * c * (1 - c)
        elif A == 9:
            if G == 0:
                y = np.where(x < -80, 0, x / (np.exp(-x) + 1))
            else:
                c = np.exp(xr)
                d = c + 1
                y = (np.where(xr > 40, x, x * c * (xr + d) / (d * d)) if G == 1 else
                     np.where(xr > 40, 0, x * c * (xr * (2 - d) + 2 * d) / (d * d * d)))
                yr = np.where(xr < -80, 0, xr / (np.exp(-xr) + 1) * gain)
    y = y * gain * dyv
    if clamp >= 0:
        if G == 0:
            y = np.where((y > -clamp) & (y < clamp), y, np.where(y >= 0, clamp, -clamp))
        else:
            y = np.where((yr > -clamp) & (yr < clamp), y, 0)
    return y


def fma(a, b, c):
    return a * b + c


def minibatch_std(x, group_size, num_channels=1):
    """Append per-group standard-deviation statistics as extra channels."""
    n, c, h, w = x.shape
    g = n if group_size is None else min(int(group_size), n)
    f = num_channels
    y = x.reshape(g, -1, f, c // f, h, w)
    y = y - y.mean(dim=0)
    y = (y.square().mean(dim=0) + 1e-8).sqrt()
    y = y.mean(dim=[2, 3, 4]).reshape(-1, f, 1, 1)
    y = y.repeat(g, 1, h, w)
    return torch.cat([x, y], dim=1)
