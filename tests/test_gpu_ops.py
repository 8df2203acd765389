"""GPU parity of the HIP kernels (through the C ABI and the op API) against the
reference's outputs (tests/golden) and the CPU oracle.  Run with -m gpu on an MI355X.

Tolerances: float32 kernels accumulate in float32 in a different order than the
CPU path -> 1e-5 relative to the tensor's max magnitude.  bf16 / fp16 kernels
accumulate in float32 and round once on store -> one unit in the last place of the
storage type (2^-8 for bf16, 2^-11 for fp16) relative to the max magnitude.
"""
import math

import numpy as np
import pytest
import torch

import golden_io
from oracle import pointwise, resample

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ULP = {torch.float32: 1e-5, torch.float64: 1e-6,  # gain and filter are float32 in the entry point's signature
       torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}


def ops():
    from ga_gan_amd.torch_utils.ops import bias_act, conv2d_resample, fma, minibatch_std, upfirdn2d
    return upfirdn2d, bias_act, conv2d_resample, fma, minibatch_std


def close(a, b, tol, what=''):
    b = b.detach().double().cpu()
    a = a.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape, what)
    scale = max(float(b.abs().max()), 1.0)
    err = float((a - b).abs().max())
    assert err <= tol * scale, (what, err, scale)


def test_library_is_loaded_from_tree():
    from ga_gan_amd import _lib
    assert _lib.lib().sg_version() >= 1000
    assert _lib.LIB_PATH.endswith('libsgops.so')


@pytest.mark.parametrize('case', golden_io.load('upfirdn2d'), ids=lambda c: str(abs(hash(str(c.meta))) % 10 ** 6))
def test_upfirdn2d_golden_fp32(case):
    upfirdn2d = ops()[0]
    m = case.meta
    x = case.t('x', DEV, grad=True)
    f = case.get('f', device=DEV)
    dy = case.t('dy', DEV, grad=True)
    y = upfirdn2d.upfirdn2d(x, f, up=m['up'], down=m['down'], padding=m['padding'], flip_filter=m['flip_filter'], gain=m['gain'])
    close(y, case.t('y'), 1e-5, 'y')
    gx, = torch.autograd.grad((y * dy).sum(), x, create_graph=True)
    close(gx, case.t('gx'), 1e-5, 'gx')
    ggdy, = torch.autograd.grad((gx * case.t('v', DEV)).sum(), dy)
    close(ggdy, case.t('ggdy'), 1e-5, 'ggdy')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float64])
@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
def test_upfirdn2d_golden_other_dtypes(dtype, layout):
    upfirdn2d = ops()[0]
    for case in golden_io.load('upfirdn2d')[::3]:
        m = case.meta
        x = case.t('x').to(dtype)
        f = case.get('f')
        want = resample.upfirdn2d(x.double(), f, up=m['up'], down=m['down'], padding=m['padding'],
                                  flip_filter=m['flip_filter'], gain=m['gain'])
        xd = x.to(DEV)
        if layout == 'nhwc':
            xd = xd.contiguous(memory_format=torch.channels_last)
        y = upfirdn2d.upfirdn2d(xd, f.to(DEV) if f is not None else None, up=m['up'], down=m['down'],
                                padding=m['padding'], flip_filter=m['flip_filter'], gain=m['gain'])
        assert y.dtype == dtype
        if layout == 'nhwc' and xd.shape[1] > 1:
            assert y.is_contiguous(memory_format=torch.channels_last)
        close(y, want, ULP[dtype], str(m))


HOT = [  # (name, x shape, up, down, padding, gain): the four instances on the generator / discriminator path
    ('blur_after_up', [4, 32, 257, 257], 1, 1, [1, 1, 1, 1], 4.0),
    ('img_skip_up', [4, 3, 128, 128], 2, 1, [2, 1, 2, 1], 4.0),
    ('pre_blur_down', [4, 32, 256, 256], 1, 1, [2, 2, 2, 2], 1.0),
    ('skip_down', [4, 32, 256, 256], 1, 2, [1, 1, 1, 1], 1.0),
    ('odd_sizes', [3, 5, 131, 77], 1, 2, [2, 1, 1, 2], 1.0),
    ('odd_up', [3, 5, 67, 45], 2, 1, [1, 2, 2, 1], 1.0),
    # rows of whole 16-byte packets on both sides: the streaming kernels (16-bit tensors)
    ('stream_blur_after_up', [3, 6, 65, 72], 1, 1, [1, -6, 1, 1], 4.0),        # several bands per wave
    ('stream_blur_wide', [1, 3, 40, 1024], 1, 1, [2, 9, 2, 2], 1.0),           # odd plane count, 129 packets: head + tail launch
    ('stream_blur_pad0', [2, 4, 33, 64], 1, 1, [0, 3, 0, 3], 1.0),
    ('stream_blur_pad3', [2, 4, 33, 64], 1, 1, [3, 0, 3, 0], 1.0),
    ('stream_down', [3, 5, 64, 128], 1, 2, [1, 1, 1, 1], 1.0),
    ('stream_down_wide', [1, 2, 70, 2064], 1, 2, [1, 1, 2, 0], 1.0),           # 129 packets out, three bands
    ('stream_down_pad3', [2, 3, 66, 80], 1, 2, [3, -1, 0, 2], 1.0),
    ('stream_up', [3, 5, 32, 64], 2, 1, [2, 1, 2, 1], 4.0),
    ('stream_up_wide', [1, 3, 35, 1032], 2, 1, [2, 1, 1, 2], 4.0),              # odd plane count, two column blocks, three bands
    ('stream_up_pad1', [2, 2, 40, 40], 2, 1, [1, 2, 3, 0], 4.0),
    ('stream_up_pad03', [2, 2, 33, 48], 2, 1, [0, 3, 0, 1], 1.0),
    ('stream_up_pad3', [2, 2, 33, 48], 2, 1, [3, 0, 2, 1], 1.0),
    # planes of 4 x 4 to 32 x 32 outputs: several planes per workgroup
    ('small_blur_16', [5, 7, 17, 24], 1, 1, [1, -6, 1, 1], 4.0),               # the filter behind a 8 -> 16 up-sampling layer
    ('small_blur_4', [3, 70, 5, 8], 1, 1, [1, -2, 1, 1], 4.0),                 # 64 planes per workgroup, a last partial one
    ('small_blur_odd', [2, 9, 11, 30], 1, 1, [2, 1, 0, 3], 1.0),               # 11 x 30 outputs
    ('small_blur_32', [2, 3, 33, 33], 1, 1, [1, 1, 1, 1], 1.0),                # one plane per workgroup
]


@pytest.mark.parametrize('name,shape,up,down,pad,gain', HOT, ids=[h[0] for h in HOT])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_upfirdn2d_hot_shapes_vs_oracle(name, shape, up, down, pad, gain, dtype):
    upfirdn2d = ops()[0]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(shape, generator=g).to(dtype)
    f = resample.make_filter([1, 3, 3, 1])
    want = resample.upfirdn2d(x.float(), f, up=up, down=down, padding=pad, gain=gain)
    xd = x.to(DEV).requires_grad_(True)
    y = upfirdn2d.upfirdn2d(xd, f.to(DEV), up=up, down=down, padding=pad, gain=gain)
    close(y, want, ULP[dtype], name)
    # tiled kernel == generic kernel (channels-last input takes the generic path)
    y2 = upfirdn2d.upfirdn2d(xd.detach().contiguous(memory_format=torch.channels_last), f.to(DEV), up=up, down=down,
                             padding=pad, gain=gain)
    close(y2, y, ULP[dtype] if dtype != torch.float32 else 1e-6, name + ' tiled-vs-generic')
    # adjoint identity <A x, v> == <x, A^T v> (backward runs the kernel with up/down swapped)
    v = torch.randn(y.shape, generator=g).to(dtype).to(DEV)
    gx, = torch.autograd.grad((y.float() * v.float()).sum(), xd)
    lhs = float((y.detach().double() * v.double()).sum())
    rhs = float((xd.detach().double() * gx.double()).sum())
    assert abs(lhs - rhs) <= (5e-2 if dtype != torch.float32 else 1e-4) * max(abs(lhs), 1.0), (lhs, rhs)


SEPARABLE = [  # (name, x shape, taps, up, down, padding, flip): one-dimensional passes of the resampler's low-pass
    ('wavelet_up', [3, 3, 141, 150], 12, 2, 1, [6, 5, 6, 5], False),
    ('wavelet_down', [3, 3, 283, 301], 12, 1, 2, [-6, -6, -6, -6], True),
    ('wavelet_down_padded', [2, 3, 130, 77], 12, 1, 2, [5, 6, 5, 6], False),
    ('seven_taps_up', [2, 5, 67, 45], 7, 2, 1, [4, 3, 3, 4], True),
    ('nine_taps_down', [2, 5, 67, 45], 9, 1, 2, [3, 4, 4, 3], False),
]


@pytest.mark.parametrize('name,shape,taps,up,down,pad,flip', SEPARABLE, ids=[h[0] for h in SEPARABLE])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_upfirdn2d_separable_vs_oracle(name, shape, taps, up, down, pad, flip, dtype):
    upfirdn2d = ops()[0]
    g = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=g).to(dtype)
    f = torch.randn([taps], generator=g)
    f = f / f.sum()
    gain = float(up * up)
    want = resample.upfirdn2d(x.double(), f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
    xd = x.to(DEV).requires_grad_(True)
    y = upfirdn2d.upfirdn2d(xd, f.to(DEV), up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
    assert y.shape == want.shape
    close(y, want, 4 * ULP[dtype], name)
    # channels-last input takes the kernel that walks the taps in global memory
    y2 = upfirdn2d.upfirdn2d(xd.detach().contiguous(memory_format=torch.channels_last), f.to(DEV), up=up, down=down,
                             padding=pad, flip_filter=flip, gain=gain)
    close(y2, y, 4 * ULP[dtype], name + ' tiled-vs-generic')
    v = torch.randn(y.shape, generator=g).to(dtype).to(DEV)
    gx, = torch.autograd.grad((y.float() * v.float()).sum(), xd)
    lhs = float((y.detach().double() * v.double()).sum())
    rhs = float((xd.detach().double() * gx.double()).sum())
    assert abs(lhs - rhs) <= (5e-2 if dtype != torch.float32 else 1e-4) * max(abs(lhs), 1.0), (lhs, rhs)


def test_upfirdn2d_linearity_full_size():
    """At the bench sizes: A(a x1 + x2) == a A(x1) + A(x2), and the constant-input response
    of the normalised [1,3,3,1] filter is the constant (interior pixels)."""
    upfirdn2d = ops()[0]
    f = resample.make_filter([1, 3, 3, 1]).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    x1 = torch.randn([8, 64, 257, 257], device=DEV, generator=g)
    x2 = torch.randn([8, 64, 257, 257], device=DEV, generator=g)
    A = lambda t: upfirdn2d.upfirdn2d(t, f, padding=[1, 1, 1, 1], gain=4.0)
    close(A(1.5 * x1 + x2), 1.5 * A(x1) + A(x2), 1e-5)
    ones = torch.ones([2, 3, 64, 64], device=DEV)
    up = upfirdn2d.upsample2d(ones, f)
    assert up.shape == (2, 3, 128, 128)
    close(up[:, :, 2:-2, 2:-2], torch.ones_like(up[:, :, 2:-2, 2:-2]), 1e-6)
    down = upfirdn2d.downsample2d(ones, f)
    assert down.shape == (2, 3, 32, 32)
    close(down[:, :, 1:-1, 1:-1], torch.ones_like(down[:, :, 1:-1, 1:-1]), 1e-6)


def test_upfirdn2d_errors():
    upfirdn2d = ops()[0]
    from ga_gan_amd import upfirdn2d_plugin
    x = torch.zeros([1, 1, 4, 4], device=DEV)
    f = torch.ones([4, 4], device=DEV)
    with pytest.raises(RuntimeError, match='output must be at least 1x1'):
        upfirdn2d_plugin.upfirdn2d(x, torch.ones([8, 8], device=DEV), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError, match='f must be float32'):
        upfirdn2d_plugin.upfirdn2d(x, f.double(), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError, match='rank 4'):
        upfirdn2d_plugin.upfirdn2d(x[0], f, 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError, match='upsampling factor'):
        upfirdn2d_plugin.upfirdn2d(x, f, 0, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError, match='CUDA'):
        upfirdn2d_plugin.upfirdn2d(x.cpu(), f, 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError):
        upfirdn2d.upfirdn2d(x.cpu(), f)
    e = upfirdn2d.upfirdn2d(torch.zeros([0, 3, 8, 8], device=DEV), f, padding=2)
    assert e.shape == (0, 3, 9, 9)


@pytest.mark.parametrize('case', golden_io.load('bias_act'), ids=lambda c: c.meta['act'])
def test_bias_act_golden_fp32(case):
    bias_act = ops()[1]
    m = case.meta
    x = case.t('x', DEV, grad=True)
    b = case.get('b', device=DEV, grad=True)
    dy = case.t('dy', DEV, grad=True)
    y = bias_act.bias_act(x, b, dim=m['dim'], act=m['act'], alpha=m['alpha'], gain=m['gain'], clamp=m['clamp'])
    close(y, case.t('y'), 1e-5, 'y')
    grads = torch.autograd.grad((y * dy).sum(), [x] + ([b] if b is not None else []), create_graph=True)
    close(grads[0], case.t('gx'), 2e-5, 'gx')
    if b is not None:
        close(grads[1], case.t('gb'), 2e-5, 'gb')
    second = torch.autograd.grad((grads[0] * case.t('v', DEV)).sum(), [dy, x], allow_unused=True)
    close(second[0], case.t('ggdy'), 2e-5, 'ggdy')
    close(second[1] if second[1] is not None else torch.zeros_like(x), case.t('ggx'), 1e-4, 'ggx')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float64])
def test_bias_act_other_dtypes_and_layouts(dtype):
    bias_act = ops()[1]
    g = torch.Generator().manual_seed(5)
    for act in ('linear', 'lrelu', 'swish'):
        for shape, dim, cl in (([4, 16, 33, 31], 1, False), ([4, 16, 8, 8], 1, True), ([6, 40], 1, False), ([3, 7, 5], 2, False)):
            x = torch.randn(shape, generator=g).to(dtype)
            b = torch.randn(shape[dim], generator=g).to(dtype)
            want = pointwise.bias_act(x.double(), b.double(), dim=dim, act=act, gain=1.5, clamp=2.0)
            xd = x.to(DEV)
            if cl:
                xd = xd.contiguous(memory_format=torch.channels_last)
            y = bias_act.bias_act(xd, b.to(DEV), dim=dim, act=act, gain=1.5, clamp=2.0)
            assert y.dtype == dtype and y.stride() == xd.stride()
            close(y, want, 2 * ULP[dtype], f'{act} {shape}')


def test_bias_act_full_size_vs_torch_fp32():
    """Bench-size tensor (32 x 64 x 256 x 256 bf16) against the same arithmetic in torch on the GPU."""
    bias_act = ops()[1]
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn([32, 64, 256, 256], device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16).requires_grad_(True)
    b = torch.randn([64], device=DEV, generator=g).to(torch.bfloat16)
    gain, clamp = math.sqrt(2), 256.0
    y = bias_act.bias_act(x, b, act='lrelu', gain=gain, clamp=clamp)
    ref = (torch.nn.functional.leaky_relu(x.float() + b.float().reshape(1, -1, 1, 1), 0.2) * gain).clamp(-clamp, clamp)
    assert float((y.float() - ref.to(torch.bfloat16).float()).abs().max()) <= 2 ** -7 * float(ref.abs().max())
    dy = torch.randn_like(y)
    gx, = torch.autograd.grad(y, x, dy)
    gref = dy.float() * torch.where(y.float() > 0, gain, gain * 0.2)
    assert float((gx.float() - gref.to(torch.bfloat16).float()).abs().max()) <= 2 ** -6 * float(gref.abs().max())


def test_bias_act_errors():
    from ga_gan_amd import bias_act_plugin
    x = torch.zeros([2, 3, 4, 4], device=DEV)
    e = torch.empty([0], device=DEV)
    with pytest.raises(RuntimeError, match='wrong number of elements'):
        bias_act_plugin.bias_act(x, torch.zeros([4], device=DEV), e, e, e, 0, 1, 3, 0.2, 1.0, -1.0)
    with pytest.raises(RuntimeError, match='same dtype'):
        bias_act_plugin.bias_act(x, torch.zeros([3], device=DEV, dtype=torch.float16), e, e, e, 0, 1, 3, 0.2, 1.0, -1.0)
    with pytest.raises(RuntimeError, match='dense'):
        bias_act_plugin.bias_act(x[:, :, ::2], e, e, e, e, 0, 1, 3, 0.2, 1.0, -1.0)
    with pytest.raises(RuntimeError, match='activation'):
        bias_act_plugin.bias_act(x, e, e, e, e, 0, 1, 12, 0.2, 1.0, -1.0)
    with pytest.raises(RuntimeError, match='CUDA'):
        bias_act_plugin.bias_act(x.cpu(), e, e, e, e, 0, 1, 3, 0.2, 1.0, -1.0)
    assert bias_act_plugin.bias_act(x[:0], e, e, e, e, 0, 1, 3, 0.2, 1.0, -1.0).shape == (0, 3, 4, 4)


@pytest.mark.parametrize('case', golden_io.load('conv2d_resample'), ids=lambda c: f"u{c.meta['up']}d{c.meta['down']}g{c.meta['groups']}")
def test_conv2d_resample_golden(case):
    conv2d_resample = ops()[2]
    m = case.meta
    x, w = case.t('x', DEV, grad=True), case.t('w', DEV, grad=True)
    y = conv2d_resample.conv2d_resample(x, w, f=case.get('f', device=DEV), up=m['up'], down=m['down'],
                                        padding=m['padding'], groups=m['groups'], flip_weight=m['flip_weight'])
    close(y, case.t('y'), 2e-5)
    gx, gw = torch.autograd.grad((y * case.t('dy', DEV)).sum(), [x, w])
    close(gx, case.t('gx'), 2e-5)
    close(gw, case.t('gw'), 2e-5)


@pytest.mark.parametrize('case', golden_io.load('misc_ops'), ids=lambda c: c.meta['op'] + str(c.meta.get('group_size')))
def test_misc_ops_golden(case):
    _, _, _, fma, minibatch_std = ops()
    m = case.meta
    if m['op'] == 'mbstd':
        x = case.t('x', DEV, grad=True)
        y = minibatch_std.minibatch_std(x, m['group_size'], m['num_channels'])
        close(y, case.t('y'), 1e-5)
        gx_plain, = torch.autograd.grad((y * case.t('dy', DEV)).sum(), x, retain_graph=True)   # backward kernel
        close(gx_plain, case.t('gx'), 1e-5)
        gx, = torch.autograd.grad((y * case.t('dy', DEV)).sum(), x, create_graph=True)        # differentiable path
        close(gx, case.t('gx'), 1e-5)
        ggx, = torch.autograd.grad((gx * case.t('v', DEV)).sum(), x)
        close(ggx, case.t('ggx'), 2e-5)
    else:
        a, b, c = case.t('a', DEV, grad=True), case.t('b', DEV, grad=True), case.t('c', DEV, grad=True)
        y = fma.fma(a, b, c)
        close(y, case.t('y'), 1e-6)
        ga, gb, gc = torch.autograd.grad((y * case.t('dy', DEV)).sum(), [a, b, c])
        close(ga, case.t('ga'), 1e-6)
        close(gb, case.t('gb'), 1e-5)
        close(gc, case.t('gc'), 1e-5)


def test_mbstd_bench_shape_vs_oracle():
    minibatch_std = ops()[4]
    g = torch.Generator().manual_seed(9)
    x = torch.randn([32, 512, 4, 4], generator=g)
    want = pointwise.minibatch_std(x, 8, 1)
    y = minibatch_std.minibatch_std(x.to(DEV), 8, 1)
    close(y, want, 1e-5)
    with pytest.raises(RuntimeError, match='divisible'):
        minibatch_std.minibatch_std(x[:6].to(DEV), 4, 1)


def test_mbstd_keeps_the_second_order_term_for_strided_inputs_and_serves_large_groups():
    """A non-contiguous input (channels-last storage) must give the gradients of the contiguous one, including the
    second-order term R1 takes through the statistic; groups beyond the kernels' 64 samples (whole-batch groups of a
    large batch) are computed as well.  Both against the CPU oracle."""
    minibatch_std = ops()[4]
    g = torch.Generator().manual_seed(10)
    x0 = torch.randn([8, 16, 4, 4], generator=g)
    dy, v = torch.randn([8, 17, 4, 4], generator=g), torch.randn([8, 16, 4, 4], generator=g)

    def run(x, fn):
        y = fn(x)
        gx, = torch.autograd.grad((y * dy.to(x.device)).sum(), x, create_graph=True)
        ggx, = torch.autograd.grad((gx * v.to(x.device)).sum(), x)
        return y, gx, ggx

    want = run(x0.clone().requires_grad_(True), lambda x: pointwise.minibatch_std(x, 4, 1))
    strided = x0.to(DEV).to(memory_format=torch.channels_last).requires_grad_(True)
    assert not strided.is_contiguous()
    got = run(strided, lambda x: minibatch_std.minibatch_std(x, 4, 1))
    for a, b in zip(got, want):
        close(a, b, 2e-5)
    assert float(want[2].abs().max()) > 0

    big = torch.randn([128, 8, 4, 4], generator=g)
    want = pointwise.minibatch_std(big, None, 1)
    close(minibatch_std.minibatch_std(big.to(DEV), None, 1), want, 1e-5)
