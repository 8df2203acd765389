"""Binding of `sg_adam_step` / `sg_lerp_multi` (include/sg_ops.h): the optimizer step of a
training phase and the generator's moving average, each as one launch per 64 tensors.

`FusedAdam` stands in for the `torch.optim.Adam(params, lr, betas, eps)` the reference's
loop constructs (training/training_loop.py:303-312) together with the gradient
sanitising it applies before every step (`param.grad.nan_to_num(0, 1e5, -1e5)`, :509).
"""
import ctypes

import torch

from . import _lib

MAX_TENSORS = 64


class AdamTensor(ctypes.Structure):
    _fields_ = [('param', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('exp_avg', ctypes.c_void_p),
                ('exp_avg_sq', ctypes.c_void_p), ('numel', ctypes.c_int64), ('step', ctypes.c_int32)]


class AdamParams(ctypes.Structure):
    _fields_ = [('tensors', ctypes.POINTER(AdamTensor)), ('count', ctypes.c_int32), ('lr', ctypes.c_float),
                ('beta1', ctypes.c_float), ('beta2', ctypes.c_float), ('eps', ctypes.c_float), ('grad_limit', ctypes.c_float)]


class LerpTensor(ctypes.Structure):
    _fields_ = [('dst', ctypes.c_void_p), ('src', ctypes.c_void_p), ('numel', ctypes.c_int64)]


class LerpParams(ctypes.Structure):
    _fields_ = [('tensors', ctypes.POINTER(LerpTensor)), ('count', ctypes.c_int32), ('beta', ctypes.c_float)]


def _check(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'{what} must reside on CUDA device')
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f'{what} must be a dense float32 tensor')


def adam_step(params, grads, exp_avgs, exp_avg_sqs, steps, lr, beta1, beta2, eps, grad_limit=1e5):
    """In-place update of `params` (and of the moment estimates; `grads` are sanitised in place).  `steps[i]` is
    the number of updates tensor i has received including this one."""
    if not (len(params) == len(grads) == len(exp_avgs) == len(exp_avg_sqs) == len(steps)):
        raise RuntimeError('adam_step: lists differ in length')
    for i in range(0, len(params), MAX_TENSORS):
        chunk = range(i, min(i + MAX_TENSORS, len(params)))
        table = (AdamTensor * len(chunk))()
        nbytes = 0
        for k, j in enumerate(chunk):
            p, g, m, v = params[j], grads[j], exp_avgs[j], exp_avg_sqs[j]
            for t, what in ((p, 'param'), (g, 'grad'), (m, 'exp_avg'), (v, 'exp_avg_sq')):
                _check(t, what)
                if t.shape != p.shape or t.device != p.device:
                    raise RuntimeError(f'adam_step: {what} {tuple(t.shape)} does not match its parameter {tuple(p.shape)}')
            table[k] = AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), int(steps[j]))
            nbytes += 7 * 4 * p.numel()   # four reads, three writes (the gradient is rewritten too: + 1)
        args = AdamParams(table, len(chunk), lr, beta1, beta2, eps, grad_limit)
        _lib.launch('sg_adam_step', args, params[i], 'adam_kernel', nbytes)
    # the kernel wrote through raw pointers: tell autograd (and every cache keyed on tensor versions)
    torch.autograd.graph.increment_version(list(params) + list(grads) + list(exp_avgs) + list(exp_avg_sqs))


def ema_(dsts, srcs, beta):
    """dst = src.lerp(dst, beta) for every pair, in place (dst the moving average, src the live weights)."""
    if len(dsts) != len(srcs):
        raise RuntimeError('ema_: lists differ in length')
    for i in range(0, len(dsts), MAX_TENSORS):
        chunk = range(i, min(i + MAX_TENSORS, len(dsts)))
        table = (LerpTensor * len(chunk))()
        nbytes = 0
        for k, j in enumerate(chunk):
            d, s = dsts[j], srcs[j]
            _check(d, 'dst')
            _check(s, 'src')
            if d.shape != s.shape or d.device != s.device:
                raise RuntimeError(f'ema_: {tuple(d.shape)} vs {tuple(s.shape)}')
            table[k] = LerpTensor(d.data_ptr(), s.data_ptr(), d.numel())
            nbytes += 3 * 4 * d.numel()
        args = LerpParams(table, len(chunk), beta)
        _lib.launch('sg_lerp_multi', args, dsts[i], 'lerp_kernel', nbytes)
    torch.autograd.graph.increment_version(list(dsts))


class FusedAdam(torch.optim.Optimizer):
    """Adam without weight decay / amsgrad whose `step()` sanitises the gradients and updates every parameter
    that has one in a single launch per 64 tensors.  State layout (`step`, `exp_avg`, `exp_avg_sq` per parameter)
    and `state_dict()` follow torch.optim.Adam, so checkpoints of one load into the other."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_limit=1e5):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, grad_limit=grad_limit))

    def load_state_dict(self, state_dict):
        """As torch.optim.Optimizer.load_state_dict; the step counts are brought back to the host afterwards (a
        snapshot read with map_location=<device> carries them as device tensors, and `step()` reads each count on
        the host: one blocking device read per parameter and step otherwise)."""
        super().load_state_dict(state_dict)
        for state in self.state.values():
            if isinstance(state.get('step'), torch.Tensor) and state['step'].device.type != 'cpu':
                state['step'] = state['step'].detach().to('cpu', torch.float32)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params, grads, ms, vs, steps = [], [], [], [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                if not state:
                    state['step'] = torch.tensor(0.0)          # (a host tensor, as torch.optim.Adam keeps it)
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state['step'] += 1
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                params.append(p)
                grads.append(p.grad)
                ms.append(state['exp_avg'])
                vs.append(state['exp_avg_sq'])
                steps.append(int(state['step']))
            if params:
                beta1, beta2 = group['betas']
                adam_step(params, grads, ms, vs, steps, group['lr'], beta1, beta2, group['eps'], group['grad_limit'])
        return loss
