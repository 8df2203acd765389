"""CPU-side checks of the C ABI: the library loads and exports every symbol that
include/sg_ops.h declares, and argument validation rejects bad calls before any
launch (no GPU needed: rejected calls never reach the device)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'sg_ops.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sg_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported():
    from ga_gan_amd import _lib
    handle = _lib.lib()
    names = declared_symbols()
    assert 'sg_upfirdn2d' in names and 'sg_bias_act' in names
    for name in names:
        assert hasattr(handle, name), name
    assert sorted(_lib.SYMBOLS) == names


def test_out_size_matches_reference_formula():
    from ga_gan_amd import _lib
    f = _lib.lib().sg_upfirdn2d_out_size
    for in_size, up, p0, p1, fs, down in [(16, 2, 2, 1, 4, 1), (33, 1, 1, 1, 4, 1), (32, 1, 1, 1, 4, 2), (9, 1, -1, 2, 12, 2)]:
        assert f(in_size, up, p0, p1, fs, down) == max((in_size * up + p0 + p1 - fs + down), 0) // down
    assert f(4, 0, 0, 0, 1, 1) == -1


def test_rejected_arguments_do_not_launch():
    from ga_gan_amd import _lib
    handle = _lib.lib()
    p = _lib.Upfirdn2dParams(up_x=0, up_y=1, down_x=1, down_y=1, f_w=1, f_h=1, out_w=1, out_h=1)
    assert handle.sg_upfirdn2d(ctypes.byref(p), None) == -1
    assert b'upsampling factor' in handle.sg_last_error()
    p = _lib.Upfirdn2dParams(up_x=1, up_y=1, down_x=1, down_y=1, f_w=1, f_h=1, out_w=0, out_h=1)
    assert handle.sg_upfirdn2d(ctypes.byref(p), None) == -1
    assert b'at least 1x1' in handle.sg_last_error()
    assert handle.sg_upfirdn2d(None, None) == -3
    b = _lib.BiasActParams(size_x=8, act=11, grad=0)
    assert handle.sg_bias_act(ctypes.byref(b), None) == -1
    b = _lib.BiasActParams(size_x=8, act=3, grad=3)
    assert handle.sg_bias_act(ctypes.byref(b), None) == -1
    b = _lib.BiasActParams(size_x=8, act=3, grad=0, dtype=0)   # NULL x
    assert handle.sg_bias_act(ctypes.byref(b), None) == -3
    m = _lib.MbstdParams(n=6, c=8, hw=16, g=4, f=1)
    assert handle.sg_mbstd_forward(ctypes.byref(m), None) == -1
    assert b'divisible' in handle.sg_last_error()


def test_product_ops_refuse_cpu_tensors():
    import torch
    from ga_gan_amd.torch_utils.ops import bias_act, upfirdn2d
    with pytest.raises(RuntimeError, match='GPU only'):
        upfirdn2d.upfirdn2d(torch.zeros(1, 1, 4, 4), None)
    with pytest.raises(RuntimeError, match='GPU only'):
        bias_act.bias_act(torch.zeros(1, 4), None)
    with pytest.raises(NotImplementedError):
        bias_act.bias_act(torch.zeros(1, 4), None, impl='ref')
