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


def test_tensors_beyond_32_bit_indexing_are_refused_with_a_message():
    """The kernels index a tensor with 32 bits.  A convolution whose input or output has 2**31 or more elements (a
    1024 x 1024 discriminator's first layer at 64 samples) is refused by name before anything is launched."""
    from ga_gan_amd import _lib, conv2d_plugin
    handle = _lib.lib()
    phase = conv2d_plugin.Phase()
    phase.nky = phase.nkx = 1
    p = conv2d_plugin.Conv2dParams(n=64, ci=3, h=1024, w_in=1024, co=32, oh=1024, ow=1024, in_step=1, out_step=1, num_phases=1,
                                   act=1, alpha=0.2, gain=1.0, clamp=-1.0, dtype=2)
    p.phase[0] = phase
    assert handle.sg_conv2d(ctypes.byref(p), None) == -1
    assert b'too large' in handle.sg_last_error()
    p.n = 63                                             # one sample fewer fits; the next complaint is the NULL pointers
    assert handle.sg_conv2d(ctypes.byref(p), None) == -3
    q = conv2d_plugin.Conv2dWgradParams(n=64, cx=3, h=1024, w_in=1024, cg=32, gh=1024, gw=1024, step=1, nky=1, nkx=1, dtype=2)
    assert handle.sg_conv2d_wgrad(ctypes.byref(q), None) == -1
    assert b'too large' in handle.sg_last_error()


def test_dmain_keeps_two_passes_when_one_would_outgrow_32_bit_indexing():
    """Both halves of Dmain share one discriminator pass only while the largest tensor of that pass stays below
    2**31 elements: at 1024 x 1024 that holds for 16 + 16 samples and not for 32 + 32."""
    import torch
    from ga_gan_amd.training import loss as loss_module
    from ga_gan_amd.training import networks
    D = networks.Discriminator(c_dim=0, img_resolution=1024, img_channels=3, channel_base=32768, channel_max=512,
                               epilogue_kwargs=dict(mbstd_group_size=4))
    assert 33554432 <= D.largest_activation_elements() < 2 ** 31 // 63
    loss = loss_module.StyleGAN2Loss(device=torch.device('cpu'), G_mapping=None, G_synthesis=None, D=D)
    assert loss._merged_groups(16, 16) == 4
    assert loss._merged_groups(32, 32) is None
    small = networks.Discriminator(c_dim=0, img_resolution=64, img_channels=3, epilogue_kwargs=dict(mbstd_group_size=4))
    assert loss_module.StyleGAN2Loss(device=torch.device('cpu'), G_mapping=None, G_synthesis=None, D=small)._merged_groups(32, 32) == 4
