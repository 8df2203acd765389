"""Oracle (test infrastructure): the optimizer step of one training phase and the generator's
moving average, on CPU in float32.

Restates:
  * gradient sanitising            DissimilarDomains/training/training_loop.py:505-510
                                   (`misc.nan_to_num(param.grad, nan=0, posinf=1e5, neginf=-1e5)`,
                                   torch_utils/misc.py:47-60)
  * the optimizer                  torch.optim.Adam as constructed at training_loop.py:303-312
                                   (`class_name='torch.optim.Adam', betas=[0, 0.99], eps=1e-8`, train.py:271-272;
                                   lazy-regularisation scaling of lr and betas at :309-311); the update rule is
                                   PyTorch's documented one: no weight decay, no amsgrad
  * the moving average             training_loop.py:516-521  (`p_ema.copy_(p.lerp(p_ema, ema_beta))`)
Pinned by running torch.optim.Adam itself - the library the reference calls - on the same tensors
(tests/test_oracle_optim.py).
"""
import math

import numpy as np


def sanitise(grad, limit=1e5):
    g = np.array(grad, dtype=np.float32, copy=True)
    g[np.isnan(g)] = 0
    g[np.isposinf(g)] = limit
    g[np.isneginf(g)] = -limit
    return g


def _lerp(start, end, weight):
    """torch.lerp's float32 arithmetic: the form that is exact at the end point the weight is nearer to."""
    f = np.float32
    w = f(weight)
    if w < 0.5:
        return (start + w * (end - start)).astype(f)
    return (end - (end - start) * (f(1) - w)).astype(f)


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, grad_limit=1e5):
    """One update; `step` counts this one.  Returns (param, sanitised grad, exp_avg, exp_avg_sq), float32."""
    f = np.float32
    g = sanitise(grad, grad_limit)
    m = _lerp(exp_avg, g, 1 - beta1)
    v = (exp_avg_sq * f(beta2) + g * g * f(1 - beta2)).astype(f)
    step_size = f(lr / (1 - beta1 ** step))
    denom = (np.sqrt(v) / f(math.sqrt(1 - beta2 ** step)) + f(eps)).astype(f)
    p = (param - step_size * (m / denom)).astype(f)
    return p, g, m, v


def ema(p_ema, p, beta):
    """p.lerp(p_ema, beta) = p + beta * (p_ema - p)."""
    return _lerp(p, p_ema, beta)
