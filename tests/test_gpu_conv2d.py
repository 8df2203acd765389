"""GPU parity of the matrix-core convolution kernel (sg_conv2d through its binding) against
torch's CPU convolution in float64 on the same (rounded) inputs.  Run with -m gpu.

Tolerance: float32 kernels use exact-float32 MFMA (an fma chain) -> 2e-5 of the output's max
magnitude; bf16 / fp16 kernels multiply 16-bit operands exactly, accumulate in float32 and round
once on store -> one unit in the last place of the storage type (2^-8 / 2^-11), doubled when the
scaled input is rounded to the storage type before the multiply (in_scale)."""
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ULP = {torch.float32: 2e-5, torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}


def close(a, b, tol, what=''):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-3)
    err = float((a - b).abs().max())
    assert err <= tol * scale, (what, err, scale)


CASES = [  # name, n, ci, co, h, w, k, kind, stride, padding
    ('3x3_same', 3, 32, 64, 16, 16, 3, 'conv', 1, 1),
    ('3x3_big', 2, 64, 64, 64, 64, 3, 'conv', 1, 1),
    ('3x3_ragged', 3, 40, 72, 19, 23, 3, 'conv', 1, 1),
    ('3x3_lowres', 5, 64, 128, 4, 4, 3, 'conv', 1, 1),
    ('3x3_narrow', 3, 32, 32, 40, 72, 3, 'conv', 1, 1),        # the weights-resident kernel: one chunk, two channel tiles
    ('3x3_narrow_ragged', 2, 24, 48, 36, 40, 3, 'conv', 1, 1),  # one chunk (partly filled), four channel tiles
    ('3x3_narrow2', 5, 56, 24, 19, 64, 3, 'conv', 1, 1),       # two chunks, two channel tiles, more tiles than workgroups would need
    ('3x3_wide', 2, 96, 80, 40, 40, 3, 'conv', 1, 1),          # the 3x3 unit-stride kernel, ragged channels and tiles
    ('3x3_wide_rect', 3, 72, 136, 36, 64, 3, 'conv', 1, 1),
    ('3x3_8x8', 9, 32, 64, 8, 8, 3, 'conv', 1, 1),
    ('1x1_torgb', 3, 64, 3, 32, 32, 1, 'conv', 1, 0),
    ('1x1_fromrgb', 3, 3, 64, 32, 32, 1, 'conv', 1, 0),
    ('1x1_skip', 2, 64, 128, 16, 16, 1, 'conv', 1, 0),
    ('3x3_stride2', 2, 32, 64, 33, 33, 3, 'conv', 2, 0),
    ('3x3_stride2_ragged', 2, 24, 48, 21, 27, 3, 'conv', 2, 0),
    ('3x3_stride2_wide', 2, 40, 72, 73, 80, 3, 'conv', 2, 0),      # the strided form of the 3x3 kernel: 36 x 39 -> not eligible (ow % 4)
    ('3x3_stride2_wide2', 2, 40, 72, 75, 88, 3, 'conv', 2, 0),     # 37 x 43 ... see below for the eligible widths
    ('3x3_stride2_fast', 3, 64, 136, 41, 72, 3, 'conv', 2, 1),     # 21 x 36 outputs, rows of whole packets: eligible
    ('3x3_stride2_fast2', 2, 96, 64, 130, 136, 3, 'conv', 2, 0),   # 64 x 67 -> not eligible; 
    ('3x3_stride2_fast3', 2, 96, 64, 67, 136, 3, 'conv', 2, 1),    # 34 x 68 outputs: eligible, ragged tiles
    ('3x3_stride2_resident', 2, 32, 64, 41, 72, 3, 'conv', 2, 1),  # one chunk, up to 64 outputs: weights stay in LDS, tiles stream
    ('3x3_stride2_resident48', 3, 32, 48, 67, 136, 3, 'conv', 2, 1),
    ('3x3_stride2_half', 3, 96, 136, 33, 40, 3, 'conv', 2, 1),       # 17 x 20 outputs: tiles of 32 columns half empty
    ('3x3_stride2_half_resident', 2, 32, 64, 33, 40, 3, 'conv', 2, 1),
    ('3x3_nopad', 2, 32, 32, 18, 18, 3, 'conv', 1, 0),
    ('t3x3_up2', 2, 64, 32, 16, 16, 3, 'conv_transpose', 2, 0),
    ('t3x3_up2_pad1', 2, 32, 32, 9, 13, 3, 'conv_transpose', 2, 1),
    ('t3x3_lowres', 4, 64, 64, 4, 4, 3, 'conv_transpose', 2, 0),
    ('t3x3_up2_wide', 2, 64, 40, 24, 24, 3, 'conv_transpose', 2, 0),    # 49 x 49 output: not a whole number of packets
    ('t3x3_up2_rect', 3, 96, 72, 19, 40, 3, 'conv_transpose', 2, 0),
]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_conv2d_plain(case, dtype):
    from ga_gan_amd import conv2d_plugin as cp
    name, n, ci, co, h, w, k, kind, stride, padding = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn([n, ci, h, w], generator=g).to(dtype)
    if kind == 'conv':
        wt = (torch.randn([co, ci, k, k], generator=g) / (ci * k * k) ** 0.5).to(dtype)
        want = F.conv2d(x.double(), wt.double(), stride=stride, padding=padding)
        plan = cp.plan_conv(k, k, stride=stride, padding=padding)
        packed = plan.pack(wt.to(DEV))
    else:
        wt = (torch.randn([ci, co, k, k], generator=g) / (ci * k * k) ** 0.5).to(dtype)
        want = F.conv_transpose2d(x.double(), wt.double(), stride=stride, padding=padding)
        plan = cp.plan_conv_transpose(k, k, stride=stride, padding=padding)
        packed = plan.pack(wt.to(DEV), transposed_layout=True)
    y = cp.conv2d(x.to(DEV), packed, plan, co)
    assert y.dtype == dtype
    close(y, want, ULP[dtype], name)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('up', [1, 2])
def test_conv2d_fused_layer(dtype, up):
    """Modulation, demodulation, noise, bias, leaky ReLU, gain and clamp around the contraction."""
    from ga_gan_amd import conv2d_plugin as cp
    g = torch.Generator().manual_seed(11 + up)
    n, ci, co, h = 4, 64, 48, 16
    x = torch.randn([n, ci, h, h], generator=g).to(dtype)
    s = torch.randn([n, ci], generator=g) * 0.5 + 1
    d = torch.rand([n, co], generator=g) + 0.5
    b = torch.randn([co], generator=g).to(dtype)
    ng = torch.tensor(0.7)
    if up == 1:
        wt = (torch.randn([co, ci, 3, 3], generator=g) / 24).to(dtype)
        plan = cp.plan_conv(3, 3, padding=1)
        packed = plan.pack(wt.to(DEV))
        conv = lambda t: F.conv2d(t, wt.double(), padding=1)
    else:
        wt = (torch.randn([ci, co, 3, 3], generator=g) / 24).to(dtype)
        plan = cp.plan_conv_transpose(3, 3, stride=2)
        packed = plan.pack(wt.to(DEV), transposed_layout=True)
        conv = lambda t: F.conv_transpose2d(t, wt.double(), stride=2)
    oh = plan.out_size(h, h)[0]
    for noise in (torch.randn([oh, oh], generator=g), torch.randn([n, 1, oh, oh], generator=g)):
        xs = (x.float() * s.reshape(n, ci, 1, 1)).to(dtype)        # the kernel rounds the scaled input to dtype
        pre = conv(xs.double()) * d.double().reshape(n, co, 1, 1) + noise.double() * 0.7 + b.double().reshape(1, co, 1, 1)
        want = (F.leaky_relu(pre, 0.2) * 1.4142).clamp(-2.0, 2.0)
        y = cp.conv2d(x.to(DEV), packed, plan, co, in_scale=s.to(DEV), out_scale=d.to(DEV), bias=b.to(DEV),
                      noise=noise.to(DEV).contiguous(), noise_gain=ng.to(DEV), act='lrelu', alpha=0.2, gain=1.4142, clamp=2.0)
        close(y, want, 2 * ULP[dtype], f'up{up} noise{tuple(noise.shape)}')


WIDE_CASES = [  # name, n, ci, co, h, w, kind: float32 layers large enough for the 512-pixel tile of the float kernel
    ('f32_wide', 8, 48, 256, 64, 64, 'conv'),
    ('f32_wide_ragged', 6, 40, 264, 50, 72, 'conv'),
    ('f32_up_33', 6, 32, 192, 32, 32, 'conv_transpose'),      # phases of 33 and 32 pixels: tiles of equal size
    ('f32_up_rect', 5, 64, 136, 35, 48, 'conv_transpose'),
    ('f32_down', 6, 40, 264, 73, 80, 'conv_s2'),                # stride 2: 128 pixel slots per workgroup
]


@pytest.mark.parametrize('case', WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
def test_conv2d_float_large_layers(case):
    """float32 layers with enough pixels for the float kernel's 512-pixel tiles, and polyphase layers whose phase
    extents are no multiple of the tile: plain, with the fused epilogue, and with the style-gradient reduction."""
    from ga_gan_amd import conv2d_plugin as cp
    name, n, ci, co, h, w, kind = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn([n, ci, h, w], generator=g)
    s = torch.randn([n, ci], generator=g) * 0.5 + 1
    d = torch.rand([n, co], generator=g) + 0.5
    b = torch.randn([co], generator=g)
    if kind == 'conv':
        wt = torch.randn([co, ci, 3, 3], generator=g) / (ci * 9) ** 0.5
        conv = lambda t: F.conv2d(t, wt.double(), padding=1)
        plan = cp.plan_conv(3, 3, padding=1)
        packed = plan.pack(wt.to(DEV))
    elif kind == 'conv_s2':
        wt = torch.randn([co, ci, 3, 3], generator=g) / (ci * 9) ** 0.5
        conv = lambda t: F.conv2d(t, wt.double(), stride=2)
        plan = cp.plan_conv(3, 3, stride=2, padding=0)
        packed = plan.pack(wt.to(DEV))
    else:
        wt = torch.randn([ci, co, 3, 3], generator=g) / (ci * 9) ** 0.5
        conv = lambda t: F.conv_transpose2d(t, wt.double(), stride=2)
        plan = cp.plan_conv_transpose(3, 3, stride=2)
        packed = plan.pack(wt.to(DEV), transposed_layout=True)
    oh, ow = plan.out_size(h, w)
    noise = torch.randn([oh, ow], generator=g)
    r = torch.randn([n, co, oh, ow], generator=g)
    xd = x.to(DEV)
    close(cp.conv2d(xd, packed, plan, co), conv(x.double()), ULP[torch.float32], name)
    c = conv(x.double() * s.double().reshape(n, ci, 1, 1))
    pre = c * d.double().reshape(n, co, 1, 1) + noise.double() * 0.7 + b.double().reshape(1, co, 1, 1)
    want = (F.leaky_relu(pre, 0.2) * 1.4142).clamp(-2.0, 2.0)
    y = cp.conv2d(xd, packed, plan, co, in_scale=s.to(DEV), out_scale=d.to(DEV), bias=b.to(DEV), noise=noise.to(DEV),
                  noise_gain=torch.tensor(0.7, device=DEV), act='lrelu', alpha=0.2, gain=1.4142, clamp=2.0)
    close(y, want, 2 * ULP[torch.float32], name + ' epilogue')
    y, red = cp.conv2d(xd, packed, plan, co, in_scale=s.to(DEV), out_scale=d.to(DEV), reduce_x=r.to(DEV))
    close(y, c * d.double().reshape(n, co, 1, 1), 2 * ULP[torch.float32], name + ' scaled')
    close(red, (c * r.double()).sum([2, 3]), 4 * ULP[torch.float32], name + ' reduction')


POINTWISE_CASES = [  # name, n, ci, co, h, w: 1x1 layers with 32 or more channels on both sides (the streaming kernel)
    ('pw_32_64', 3, 32, 64, 64, 64),
    ('pw_64_32', 2, 64, 32, 48, 40),            # a last tile of 128 pixels
    ('pw_64_128', 5, 64, 128, 40, 40),          # 1600 pixels: ragged last tile; more tiles than a multiple of eight
    ('pw_96_48', 2, 96, 48, 32, 36),
    ('pw_128_200', 2, 128, 200, 36, 32),        # two channel blocks, the second partly filled
    ('pw_256_512', 1, 256, 512, 32, 32),        # eight chunks, four channel blocks
]


@pytest.mark.parametrize('case', POINTWISE_CASES, ids=[c[0] for c in POINTWISE_CASES])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_conv2d_pointwise(case, dtype):
    from ga_gan_amd import conv2d_plugin as cp
    name, n, ci, co, h, w = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn([n, ci, h, w], generator=g).to(dtype)
    wt = (torch.randn([co, ci, 1, 1], generator=g) / ci ** 0.5).to(dtype)
    b = torch.randn([co], generator=g).to(dtype)
    plan = cp.plan_conv(1, 1)
    packed = plan.pack(wt.to(DEV))
    assert cp.kernel_name(x.to(DEV), plan, co) == 'pointwise_kernel'
    c = F.conv2d(x.double(), wt.double())
    close(cp.conv2d(x.to(DEV), packed, plan, co, gain=0.5 ** 0.5), c * 0.5 ** 0.5, ULP[dtype], name)
    want = (F.leaky_relu(c + b.double().reshape(1, co, 1, 1), 0.2) * 1.4142).clamp(-1.5, 1.5)
    y = cp.conv2d(x.to(DEV), packed, plan, co, bias=b.to(DEV), act='lrelu', alpha=0.2, gain=1.4142, clamp=1.5)
    close(y, want, 2 * ULP[dtype], name + ' epilogue')
    # the data gradient of such a layer is the same kernel with the weights transposed
    plan_t = cp.plan_conv_transpose(1, 1, stride=1)
    gy = torch.randn([n, co, h, w], generator=g).to(dtype)
    dx = cp.conv2d(gy.to(DEV), plan_t.pack(wt.to(DEV), transposed_layout=True), plan_t, ci)
    close(dx, F.conv_transpose2d(gy.double(), wt.double()), ULP[dtype], name + ' transposed')


def test_conv2d_bench_shapes_match_library_conv():
    """Bench-size layers (256-config, batch 8) against the library convolution on the GPU in float32."""
    from ga_gan_amd import conv2d_plugin as cp
    g = torch.Generator(device=DEV).manual_seed(5)
    for ci, co, h in ((64, 64, 256), (128, 128, 128), (512, 512, 32)):
        x = torch.randn([8, ci, h, h], device=DEV, generator=g).to(torch.bfloat16)
        wt = (torch.randn([co, ci, 3, 3], device=DEV, generator=g) / (ci * 9) ** 0.5).to(torch.bfloat16)
        plan = cp.plan_conv(3, 3, padding=1)
        y = cp.conv2d(x, plan.pack(wt), plan, co)
        want = F.conv2d(x.float(), wt.float(), padding=1)
        close(y, want, 2 ** -7, f'{ci}->{co}@{h}')


def test_conv2d_errors():
    from ga_gan_amd import conv2d_plugin as cp
    plan = cp.plan_conv(3, 3, padding=1)
    x = torch.zeros([1, 8, 8, 8], device=DEV)
    w = plan.pack(torch.zeros([8, 8, 3, 3], device=DEV))
    with pytest.raises(RuntimeError, match='CUDA'):
        cp.conv2d(x.cpu(), w, plan, 8)
    with pytest.raises(RuntimeError, match='dtype'):
        cp.conv2d(x, w.half(), plan, 8)
    with pytest.raises(RuntimeError, match='elements'):
        cp.conv2d(x, w[:-1].contiguous(), plan, 8)
    with pytest.raises(RuntimeError, match='in_scale'):
        cp.conv2d(x, w, plan, 8, in_scale=torch.zeros([1, 4], device=DEV))
    with pytest.raises((RuntimeError, KeyError)):
        cp.conv2d(x.double(), w.double(), plan, 8)
    assert cp.conv2d(x[:0], w, plan, 8).shape == (0, 8, 8, 8)


WGRAD_CASES = [  # name, n, ci, co, h, w, k, stride, padding
    ('3x3_same', 3, 32, 64, 16, 16, 3, 1, 1),
    ('3x3_big', 4, 64, 64, 64, 64, 3, 1, 1),
    ('3x3_ragged', 3, 40, 72, 19, 23, 3, 1, 1),
    ('3x3_wide_ragged', 3, 40, 72, 36, 48, 3, 1, 1),           # the 3x3 unit-stride weight-gradient kernel
    ('3x3_lowres', 8, 64, 96, 4, 4, 3, 1, 1),
    ('1x1', 3, 64, 3, 32, 32, 1, 1, 0),
    ('1x1_fromrgb', 3, 3, 64, 32, 32, 1, 1, 0),
    ('3x3_stride2', 2, 32, 64, 33, 33, 3, 2, 0),
    ('3x3_nopad', 2, 32, 32, 18, 18, 3, 1, 0),
    ('3x3_lowres_wide', 5, 136, 72, 16, 16, 3, 1, 1),          # float32: 64 x 64 blocks, ragged channel blocks
    ('3x3_stride2_lowres', 9, 64, 80, 17, 24, 3, 2, 0),
    ('3x3_odd_width', 3, 48, 40, 9, 11, 3, 1, 1),               # rows that are no whole packets: scalar staging
    ('1x1_lowres', 6, 96, 64, 8, 8, 1, 1, 0),
    # 64 x 64 blocks with at most 32 real channels on one or both sides: waves share the k-steps of the real sub-block
    ('3x3_32_32', 2, 32, 32, 32, 64, 3, 1, 1),
    ('3x3_32_64', 2, 32, 64, 32, 32, 3, 1, 1),
    ('3x3_64_32', 2, 64, 32, 24, 32, 3, 1, 1),
    ('3x3_96_96', 2, 96, 96, 16, 32, 3, 1, 1),
    ('3x3_stride2_32_32', 2, 32, 32, 16, 64, 3, 2, 1),
    ('3x3_stride2_32_96', 2, 32, 96, 24, 64, 3, 2, 1),
    ('3x3_stride2_g16', 5, 64, 96, 40, 40, 3, 2, 0),            # gradient planes 19 x 19 / 16 columns wide: half-filled tiles
    ('3x3_stride2_g16b', 4, 64, 64, 33, 40, 3, 2, 0),
    ('3x3_g16', 3, 64, 64, 12, 16, 3, 1, 1),
    # 16-bit 1x1 layers with 32 or more channels on both sides: blocks of 128 x 64 weights
    ('1x1_skip', 3, 64, 128, 32, 32, 1, 1, 0),
    ('1x1_skip_ragged', 2, 72, 200, 36, 40, 1, 1, 0),           # partly filled blocks, a last chunk of 32 pixels
    ('1x1_skip_wide', 2, 256, 136, 16, 24, 1, 1, 0),            # four block columns, two block rows
]


@pytest.mark.parametrize('case', WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_conv2d_wgrad(case, dtype):
    from ga_gan_amd import conv2d_plugin as cp
    name, n, ci, co, h, w, k, stride, padding = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn([n, ci, h, w], generator=g).to(dtype)
    wt = torch.zeros([co, ci, k, k], dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), wt, stride=stride, padding=padding)
    dy = torch.randn(y.shape, generator=g).to(dtype)
    want, = torch.autograd.grad(y, wt, dy.double())
    xs, gs = torch.rand([n, ci], generator=g) + 0.5, torch.rand([n, co], generator=g) + 0.5
    dw = cp.conv2d_wgrad(x.to(DEV), dy.to(DEV), k, k, stride=stride, padding=padding)
    assert dw.dtype == torch.float32 and dw.shape == wt.shape
    # products of 16-bit operands are exact, sums are float32: only summation-order error is left
    close(dw, want, 2e-5 if dtype == torch.float32 else 1e-4, name)
    # with per-sample scales (operands are rounded to dtype after scaling)
    x2 = (x.float() * xs.reshape(n, ci, 1, 1)).to(dtype)
    dy2 = (dy.float() * gs.reshape(n, co, 1, 1)).to(dtype)
    want2, = torch.autograd.grad(F.conv2d(x2.double(), wt, stride=stride, padding=padding), wt, dy2.double())
    dw2 = cp.conv2d_wgrad(x.to(DEV), dy.to(DEV), k, k, stride=stride, padding=padding, x_scale=xs.to(DEV), g_scale=gs.to(DEV))
    # (a scaled operand that lands on a rounding tie may round the other way on the device: one unit in its last place)
    close(dw2, want2, 2e-5 if dtype == torch.float32 else 4e-4, name + ' scaled')


def test_conv2d_wgrad_transposed_roles():
    """Weight gradient of a transposed convolution = the same kernel with x and dy exchanged."""
    from ga_gan_amd import conv2d_plugin as cp
    g = torch.Generator().manual_seed(3)
    x = torch.randn([2, 32, 9, 9], generator=g).to(torch.bfloat16)
    wt = torch.zeros([32, 48, 3, 3], dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(x.double(), wt, stride=2)
    dy = torch.randn(y.shape, generator=g).to(torch.bfloat16)
    want, = torch.autograd.grad(y, wt, dy.double())
    dw = cp.conv2d_wgrad(dy.to(DEV), x.to(DEV), 3, 3, stride=2, padding=0)
    close(dw, want, 1e-4)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 40, 72, 41, 72, 20, 32), (3, 64, 136, 23, 136, 11, 64), (2, 96, 64, 9, 80, 4, 40)],
                         ids=['40to72', '64to136', '96to64'])
def test_conv2d_wgrad_stride2_cropped(shape, dtype):
    """Stride-2 weight gradient with a gradient plane narrower than the convolution's natural output (rows of
    whole packets, as the down-sampling layers produce them): the 3x3 strided weight-gradient kernel."""
    from ga_gan_amd import conv2d_plugin as cp
    n, ci, co, h, w, gh, gw = shape
    g = torch.Generator().manual_seed(ci + gw)
    x = torch.randn([n, ci, h, w], generator=g).to(dtype)
    wt = torch.zeros([co, ci, 3, 3], dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), wt, stride=2, padding=1)
    assert y.shape[2] >= gh and y.shape[3] >= gw
    dy = torch.randn([n, co, gh, gw], generator=g).to(dtype)
    dy_full = torch.zeros(y.shape, dtype=torch.float64)
    dy_full[:, :, :gh, :gw] = dy.double()
    want, = torch.autograd.grad(y, wt, dy_full)
    xs, gs = torch.rand([n, ci], generator=g) + 0.5, torch.rand([n, co], generator=g) + 0.5
    dw = cp.conv2d_wgrad(x.to(DEV), dy.to(DEV), 3, 3, stride=2, padding=1)
    close(dw, want, 1e-4, 'plain')
    x2 = (x.float() * xs.reshape(n, ci, 1, 1)).to(dtype)
    dy2 = torch.zeros(y.shape, dtype=torch.float64)
    dy2[:, :, :gh, :gw] = (dy.float() * gs.reshape(n, co, 1, 1)).to(dtype).double()
    want2, = torch.autograd.grad(F.conv2d(x2.double(), wt, stride=2, padding=1), wt, dy2)
    dw2 = cp.conv2d_wgrad(x.to(DEV), dy.to(DEV), 3, 3, stride=2, padding=1, x_scale=xs.to(DEV), g_scale=gs.to(DEV))
    close(dw2, want2, 1e-4, 'scaled')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 64, 40, 24, 24), (3, 96, 72, 19, 40), (2, 32, 8, 33, 16), (1, 128, 64, 8, 64)],
                         ids=['64to40', '96to72', '32to8', '128to64'])
def test_conv2d_transposed_padded_rows(shape, dtype):
    """The up-sampling call of the training path: 3x3 stride-2 transposed convolution with modulated input whose
    output rows are padded to whole 16-byte packets (zero columns): the one-contraction up-sampling kernel."""
    from ga_gan_amd import conv2d_plugin as cp
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(ci * 7 + w)
    x = torch.randn([n, ci, h, w], generator=g).to(dtype)
    wt = (torch.randn([ci, co, 3, 3], generator=g) / (ci * 9) ** 0.5).to(dtype)
    s = torch.rand([n, ci], generator=g) + 0.5
    plan = cp.plan_conv_transpose(3, 3, stride=2)
    oh, ow = plan.out_size(h, w)
    ow_pad = (ow + 7) // 8 * 8
    y = cp.conv2d(x.to(DEV), plan.pack(wt.to(DEV), transposed_layout=True), plan, co, in_scale=s.to(DEV), out_hw=(oh, ow_pad))
    xs = (x.float() * s.reshape(n, ci, 1, 1)).to(dtype)
    want = F.conv_transpose2d(xs.double(), wt.double(), stride=2)
    assert y.shape == (n, co, oh, ow_pad)
    close(y[..., :ow], want, 2 * ULP[dtype], 'valid part')
    assert float(y[..., ow:].abs().max()) == 0, 'padding columns must be zero'


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_pack_weights_kernel_matches_composition(dtype):
    """sg_pack_weights against the same packing composed from slices, flips, permuting copies and a concatenation,
    for plain, strided and polyphase-transposed plans, dense and transposed weight views: bit-exact."""
    from ga_gan_amd import conv2d_plugin as cp
    g = torch.Generator().manual_seed(3)
    for co, ci, k in ((5, 7, 3), (64, 32, 3), (3, 16, 1), (8, 8, 5)):
        w = torch.randn(co, ci, k, k, generator=g).to(DEV).to(dtype)
        plans = [cp.plan_conv(k, k, padding=k // 2), cp.plan_conv(k, k, stride=2, padding=0)]
        if k > 1:
            plans.append(cp.plan_conv_transpose(k, k, stride=2, padding=0))
        for plan in plans:
            for transposed in (False, True):
                got = plan._pack(w, transposed)
                src = w.transpose(0, 1) if transposed else w
                parts = []
                for (_, _, _, _, nky, nkx, ky_idx, kx_idx) in plan.phases:
                    sel = src[:, :, list(ky_idx)][:, :, :, list(kx_idx)]
                    parts.append(sel.permute(2, 3, 0, 1).reshape(-1))
                want = torch.cat(parts)
                assert got.dtype == dtype and got.shape == want.shape
                assert torch.equal(got, want), (co, ci, k, plan.kind, transposed)


def test_float32_prepared_weights_give_the_same_bits():
    """A float32 call whose packed weight is a constant keeps the weight's largest magnitude and split rows with it
    (sg_conv2d_prepare_weights); the result equals the call that measures and splits the weight itself, bit for bit,
    for plain, strided and transposed geometries, and a second call serves the kept buffer."""
    from ga_gan_amd import conv2d_plugin as cp
    from ga_gan_amd.torch_utils import memo
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 96, 20, 24, generator=g).to(DEV)
    s = (torch.rand(4, 96, generator=g) + 0.5).to(DEV)
    for plan, co in ((cp.plan_conv(3, 3, padding=1), 80), (cp.plan_conv(3, 3, stride=2, padding=0), 64),
                     (cp.plan_conv_transpose(3, 3, stride=2, padding=0), 48), (cp.plan_conv(1, 1), 3)):
        shape = [96, co, 3, 3] if plan.kind == 'conv_transpose' else [co, 96, plan.kh, plan.kw]
        w = (torch.randn(shape, generator=g) * 0.05).to(DEV)
        packed = plan._pack(w, plan.kind == 'conv_transpose')
        plain = cp.conv2d(x, packed, plan, co, in_scale=s)
        kept = memo.constant(packed.clone())
        first = cp.conv2d(x, kept, plan, co, in_scale=s)
        assert any(k[0] == 'float32 prepared' for k in kept._sg_memo), plan.kind
        again = cp.conv2d(x, kept, plan, co, in_scale=s)
        assert torch.equal(first, plain) and torch.equal(again, plain), plan.kind
