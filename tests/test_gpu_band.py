"""GPU parity of the row-band 3x3 kernel (same-size convolution, 32 input channels, 32 or 64 output channels,
16-bit) through `sg_conv2d`, against float64 arithmetic on the host.  The kernel folds the per-sample input
scale into the weights (w * s rounded to the storage type), so the expectation does the same.  Tolerance: products
of 16-bit operands are exact and sums are float32, so what is left is the rounding of the result (one unit in
the last place of the storage type: 2^-8 bf16, 2^-11 fp16, doubled for the epilogue arithmetic).  Run with -m gpu."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ULP = {torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}

SHAPES = [  # n, co, h, w
    (2, 32, 40, 72),      # ragged rows (5 bands of 8), two steps, the second partly filled
    (3, 64, 19, 64),      # two channel groups, last band has 3 rows
    (2, 32, 16, 200),     # four steps, ragged last step
    (1, 32, 8, 64),       # the smallest eligible image
    (5, 64, 64, 128),     # more items than fit one round of a small grid
]


def close(a, b, tol, what=''):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-3)
    err = float((a - b).abs().max())
    assert err <= tol * scale, (what, err, scale)


def kernel_name(cp, x, packed, plan, co, **kw):
    from ga_gan_amd import _lib
    n, ci, h, w = x.shape
    p = cp.Conv2dParams(x=_lib.ptr(x), w=_lib.ptr(packed), y=_lib.ptr(x), dtype=_lib.dtype_code(x), n=n, ci=ci, h=h, w_in=w,
                        co=co, oh=h, ow=w, in_step=1, out_step=1, num_phases=1, act=1, gain=1.0, clamp=-1.0, mask_clamp=-1.0)
    p.phase[0] = cp.Phase(0, 0, -1, -1, 3, 3, 0)
    return _lib.lib().sg_conv2d_kernel(ctypes.byref(p)).decode()


def modulated_conv(x, wt, s):
    """conv(x[n], round(w * s[n])) per sample in float64; x, wt in the storage dtype, s float32 [n, ci] or None."""
    outs = []
    for i in range(x.shape[0]):
        w_i = wt.float() if s is None else (wt.float() * s[i].reshape(1, -1, 1, 1)).to(wt.dtype).float()
        outs.append(F.conv2d(x[i:i + 1].double(), w_i.double(), padding=1))
    return torch.cat(outs)


@pytest.mark.parametrize('shape', SHAPES, ids=[f'n{s[0]}_co{s[1]}_{s[2]}x{s[3]}' for s in SHAPES])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_band_plain_and_modulated(shape, dtype):
    from ga_gan_amd import conv2d_plugin as cp
    n, co, h, w = shape
    g = torch.Generator().manual_seed(n * 1000 + co + h + w)
    x = torch.randn([n, 32, h, w], generator=g).to(dtype)
    wt = (torch.randn([co, 32, 3, 3], generator=g) / 17).to(dtype)
    s = torch.rand([n, 32], generator=g) + 0.5
    plan = cp.plan_conv(3, 3, padding=1)
    packed = plan.pack(wt.to(DEV))
    assert kernel_name(cp, x.to(DEV), packed, plan, co) == 'band3x3_kernel'
    close(cp.conv2d(x.to(DEV), packed, plan, co), modulated_conv(x, wt, None), ULP[dtype], 'plain')
    close(cp.conv2d(x.to(DEV), packed, plan, co, in_scale=s.to(DEV)), modulated_conv(x, wt, s), ULP[dtype], 'modulated')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('per_sample_noise', [False, True])
def test_band_layer_epilogue(dtype, per_sample_noise):
    """Demodulation, noise (one plane or one per sample), bias, leaky ReLU, gain and clamp behind the contraction."""
    from ga_gan_amd import conv2d_plugin as cp
    n, co, h, w = 3, 64, 27, 136
    g = torch.Generator().manual_seed(77)
    x = torch.randn([n, 32, h, w], generator=g).to(dtype)
    wt = (torch.randn([co, 32, 3, 3], generator=g) / 17).to(dtype)
    s = torch.rand([n, 32], generator=g) + 0.5
    d = torch.rand([n, co], generator=g) + 0.5
    b = torch.randn([co], generator=g).to(dtype)
    noise = torch.randn([n, 1, h, w] if per_sample_noise else [h, w], generator=g)
    plan = cp.plan_conv(3, 3, padding=1)
    y = cp.conv2d(x.to(DEV), plan.pack(wt.to(DEV)), plan, co, in_scale=s.to(DEV), out_scale=d.to(DEV), bias=b.to(DEV),
                  noise=noise.to(DEV).contiguous(), noise_gain=torch.tensor(0.7, device=DEV), act='lrelu', alpha=0.2, gain=1.4142,
                  clamp=2.0)
    pre = modulated_conv(x, wt, s) * d.double().reshape(n, co, 1, 1) + noise.double() * 0.7 + b.double().reshape(1, co, 1, 1)
    want = (F.leaky_relu(pre, 0.2) * 1.4142).clamp(-2.0, 2.0)
    close(y, want, 2 * ULP[dtype])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('masked', [False, True])
@pytest.mark.parametrize('reduce', [False, True])
def test_band_data_gradient(dtype, masked, reduce):
    """The call the backward pass makes: incoming gradient times the activation mask of the saved output while it is
    staged, per-sample scales on both sides, the style-gradient reduction as a by-product."""
    from ga_gan_amd import conv2d_plugin as cp
    n, co, h, w = 3, 32, 26, 136
    g = torch.Generator().manual_seed(5)
    dy = torch.randn([n, 32, h, w], generator=g).to(dtype)
    saved = (torch.randn([n, 32, h, w], generator=g) * 1.2).to(dtype)      # some beyond the clamp bound of 2
    saved[0, 0, 0, :8] = 0                                                 # zero counts as the negative side
    wt = (torch.randn([co, 32, 3, 3], generator=g) / 17).to(dtype)
    d = torch.rand([n, 32], generator=g) + 0.5
    s = torch.rand([n, co], generator=g) + 0.5
    rx = torch.randn([n, co, h, w], generator=g).to(dtype)
    plan = cp.plan_conv(3, 3, padding=1)
    kw = dict(mask_y=saved.to(DEV), mask_alpha=0.2, mask_clamp=2.0) if masked else {}
    if masked:
        assert cp.supports_mask(dy.to(DEV), plan, co, reduce=reduce)
    out = cp.conv2d(dy.to(DEV), plan.pack(wt.to(DEV)), plan, co, in_scale=d.to(DEV), out_scale=s.to(DEV), gain=1.4142,
                    reduce_x=rx.to(DEV) if reduce else None, **kw)
    gin = dy.float()
    if masked:
        m = torch.where(saved.float() > 0, 1.0, 0.2) * (saved.float().abs() < 2.0)
        gin = (gin * m).to(dtype).float()
    acc = modulated_conv(gin.to(dtype), wt, d)
    want = acc * s.double().reshape(n, co, 1, 1) * 1.4142
    if reduce:
        out, r = out
        want_r = (acc * rx.double()).sum(dim=(2, 3))
        close(r, want_r, 2e-3, 'reduction')     # bf16 products summed in float32 against float64
    close(out, want, 2 * ULP[dtype])


def test_mask_is_refused_where_no_kernel_applies_it():
    from ga_gan_amd import conv2d_plugin as cp
    x = torch.zeros([1, 48, 16, 64], device=DEV, dtype=torch.bfloat16)
    plan = cp.plan_conv(3, 3, padding=1)
    assert not cp.supports_mask(x, plan, 48)
    with pytest.raises(RuntimeError, match='input mask'):
        cp.conv2d(x, plan.pack(torch.zeros([48, 48, 3, 3], device=DEV, dtype=torch.bfloat16)), plan, 48, mask_y=x)


ROWS_CASES = [
    # n, cx, cg, h, w, masked, scales
    (2, 32, 32, 64, 64, False, True),
    (2, 32, 32, 64, 128, True, True),
    (1, 32, 32, 32, 64, True, False),
    (3, 32, 32, 16, 64, False, False),
    (2, 32, 32, 8, 192, True, True),
    (33, 32, 32, 4, 64, True, True),      # more tiles than one XCD's share of workgroups, ragged shares
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('case', ROWS_CASES, ids=[f'r{i}' for i in range(len(ROWS_CASES))])
def test_weight_gradient_of_narrow_same_size_layers(case, dtype):
    """wgrad_rows_kernel (3x3, unit stride, 32 channels, rows of whole 64-pixel tiles) with and without the
    activation mask on the gradient operand, against float64 arithmetic on the operands as the kernel rounds them
    (g * scale * mask and x * scale, to the 16-bit type, as the kernels it replaces do): 1e-5 relative L2, the float32
    accumulation."""
    from ga_gan_amd import conv2d_plugin as cp
    n, cx, cg, h, w, masked, scales = case
    gen = torch.Generator(device='cuda').manual_seed(h * 7 + w)
    x = torch.randn([n, cx, h, w], device='cuda', generator=gen).to(dtype)
    g = torch.randn([n, cg, h, w], device='cuda', generator=gen).to(dtype)
    y = (torch.randn([n, cg, h, w], device='cuda', generator=gen) * 2).to(dtype)
    xs = (1 + 0.3 * torch.randn([n, cx], device='cuda', generator=gen)) if scales else None
    gs = (1 + 0.3 * torch.rand([n, cg], device='cuda', generator=gen)) if scales else None
    p = cp._wgrad_params(x, g, 3, 3, 1, 1, xs, gs, None, y if masked else None, 0.2, 1.5)
    import ctypes
    from ga_gan_amd import _lib
    assert _lib.lib().sg_conv2d_wgrad_kernel(ctypes.byref(p)) == b'wgrad_rows_kernel'
    assert cp.wgrad_supports_mask(x, g, 3, 3, 1, 1)
    kw = dict(mask_y=y, mask_alpha=0.2, mask_clamp=1.5) if masked else {}
    dw = cp.conv2d_wgrad(x, g, 3, 3, stride=1, padding=1, x_scale=xs, g_scale=gs, **kw)
    gf, xf = g.float(), x.float()                 # the kernel's own order: scale, mask, round to the 16-bit type
    if scales:
        gf = gf * gs.reshape(n, cg, 1, 1)
        xf = xf * xs.reshape(n, cx, 1, 1)
    if masked:
        gf = gf * torch.where(y > 0, 1.0, 0.2).float() * (y.abs() < 1.5).float()
    gd, xd = gf.to(dtype).double(), xf.to(dtype).double()
    want = torch.nn.grad.conv2d_weight(xd.cpu(), (cg, cx, 3, 3), gd.cpu(), padding=1)
    err = float((dw.double().cpu() - want).norm() / want.norm())
    assert err < 1e-5, err
