"""2-D convolution entry points of the op API (reference: torch_utils/ops/conv2d_gradfix.py).

`conv2d` / `conv_transpose2d` dispatch to the library convolution of PyTorch-ROCm
(MIOpen), whose autograd supports gradients of any order.  `no_weight_gradients()`
is kept for callers that wrap their regularisation passes in it
(training/loss.py:98,142 of the reference); with `torch.autograd.grad(inputs=...)`
weight gradients that nobody asked for are not computed in the first place.
"""
import contextlib

import torch

enabled = False
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                      dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)
