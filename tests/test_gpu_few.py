"""GPU parity of the streaming 1x1 kernels for layers with at most 4 channels on one side (to-RGB, from-RGB, their
data and weight gradients) through `sg_conv2d` / `sg_conv2d_wgrad`, against float64 arithmetic on the host.  The
kernels multiply float32 copies of the 16-bit operands (weights times the per-sample scale stay float32) and round
once on store: one unit in the last place of the storage type, doubled for the epilogue.  Run with -m gpu."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ULP = {torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}


def close(a, b, tol, what=''):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-3)
    err = float((a - b).abs().max())
    assert err <= tol * scale, (what, err, scale)


def kernel_name(cp, x, co):
    from ga_gan_amd import _lib
    n, ci, h, w = x.shape
    p = cp.Conv2dParams(x=_lib.ptr(x), w=_lib.ptr(x), y=_lib.ptr(x), dtype=_lib.dtype_code(x), n=n, ci=ci, h=h, w_in=w, co=co,
                        oh=h, ow=w, in_step=1, out_step=1, num_phases=1, act=1, gain=1.0, clamp=-1.0, mask_clamp=-1.0)
    p.phase[0] = cp.Phase(0, 0, 0, 0, 1, 1, 0)
    return _lib.lib().sg_conv2d_kernel(ctypes.byref(p)).decode()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 32, 3, 64, 64), (3, 72, 3, 48, 96), (2, 512, 4, 64, 72), (1, 20, 1, 64, 64)],
                         ids=['32to3', '72to3', '512to4', '20to1'])
def test_to_rgb_forward(shape, dtype):
    from ga_gan_amd import conv2d_plugin as cp
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(ci + co)
    x = torch.randn([n, ci, h, w], generator=g).to(dtype)
    wt = (torch.randn([co, ci, 1, 1], generator=g) / ci ** 0.5).to(dtype)
    s = torch.rand([n, ci], generator=g) + 0.5
    b = torch.randn([co], generator=g).to(dtype)
    plan = cp.plan_conv(1, 1)
    assert kernel_name(cp, x.to(DEV), co) == 'conv1x1_few_kernel'
    y = cp.conv2d(x.to(DEV), plan.pack(wt.to(DEV)), plan, co, in_scale=s.to(DEV), bias=b.to(DEV), clamp=1.5)
    want = torch.einsum('nchw,oc,nc->nohw', x.double(), wt.double()[:, :, 0, 0], s.double()) + b.double().reshape(1, co, 1, 1)
    close(y, want.clamp(-1.5, 1.5), 2 * ULP[dtype])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 3, 32, 64, 64), (3, 3, 72, 48, 96), (2, 4, 64, 64, 72), (1, 1, 16, 64, 64)],
                         ids=['3to32', '3to72', '4to64', '1to16'])
def test_from_rgb_forward(shape, dtype):
    from ga_gan_amd import conv2d_plugin as cp
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(ci + co)
    x = torch.randn([n, ci, h, w], generator=g).to(dtype)
    wt = (torch.randn([co, ci, 1, 1], generator=g) / ci ** 0.5).to(dtype)
    b = torch.randn([co], generator=g).to(dtype)
    d = torch.rand([n, co], generator=g) + 0.5
    plan = cp.plan_conv(1, 1)
    assert kernel_name(cp, x.to(DEV), co) == 'conv1x1_few_kernel'
    y = cp.conv2d(x.to(DEV), plan.pack(wt.to(DEV)), plan, co, out_scale=d.to(DEV), bias=b.to(DEV), act='lrelu', alpha=0.2,
                  gain=1.4142, clamp=2.0)
    pre = torch.einsum('nchw,oc->nohw', x.double(), wt.double()[:, :, 0, 0]) * d.double().reshape(n, co, 1, 1)
    want = (F.leaky_relu(pre + b.double().reshape(1, co, 1, 1), 0.2) * 1.4142).clamp(-2.0, 2.0)
    close(y, want, 2 * ULP[dtype])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('co', [32, 64, 48, 128])
@pytest.mark.parametrize('masked', [False, True])
def test_to_rgb_data_gradient(dtype, co, masked):
    """dx[n, c, p] = s[n, c] * gain * sum_k w[k, c] (dy * m)[n, k, p], with sum_p acc * x as a by-product."""
    from ga_gan_amd import conv2d_plugin as cp
    n, k, h, w = 3, 3, 64, 80
    reduce = True      # channel counts other than 32 / 64 take the general kernel, whose partial rows differ
    if masked and co not in (32, 64):
        pytest.skip('no masking kernel for this geometry with a reduction')
    g = torch.Generator().manual_seed(co)
    dy = torch.randn([n, k, h, w], generator=g).to(dtype)
    saved = (torch.randn([n, k, h, w], generator=g) * 1.2).to(dtype)
    wt = (torch.randn([co, k, 1, 1], generator=g)).to(dtype)      # the transposed weight: [out = layer input channels, in = 3]
    s = torch.rand([n, co], generator=g) + 0.5
    rx = torch.randn([n, co, h, w], generator=g).to(dtype)
    plan = cp.plan_conv(1, 1)
    kw = dict(mask_y=saved.to(DEV), mask_alpha=1.0, mask_clamp=2.0) if masked else {}
    if masked:
        assert cp.supports_mask(dy.to(DEV), plan, co, reduce=reduce)
    out = cp.conv2d(dy.to(DEV), plan.pack(wt.to(DEV)), plan, co, out_scale=s.to(DEV), gain=0.5,
                    reduce_x=rx.to(DEV) if reduce else None, **kw)
    gin = dy.double()
    if masked:
        gin = gin * (saved.double().abs() < 2.0)
    acc = torch.einsum('nkhw,ok->nohw', gin, wt.double()[:, :, 0, 0])
    if reduce:
        out, r = out
        close(r, (acc * rx.double()).sum(dim=(2, 3)), 1e-4, 'reduction')
    close(out, acc * s.double().reshape(n, co, 1, 1) * 0.5, 2 * ULP[dtype])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('co', [32, 64, 48, 128])
@pytest.mark.parametrize('reduce', [False, True])
@pytest.mark.parametrize('masked', [False, True])
def test_to_rgb_data_gradient_with_addend(dtype, co, reduce, masked):
    """The same call with the other gradient of the layer's input as `addend`: bit-equal to the call without it followed
    by a sum of the two 16-bit tensors, and the reduction untouched."""
    from ga_gan_amd import conv2d_plugin as cp
    n, k, h, w = 3, 3, 64, 80
    g = torch.Generator().manual_seed(co + 1000)
    dy = torch.randn([n, k, h, w], generator=g).to(dtype).to(DEV)
    saved = (torch.randn([n, k, h, w], generator=g) * 1.2).to(dtype).to(DEV)
    wt = (torch.randn([co, k, 1, 1], generator=g)).to(dtype).to(DEV)
    s = (torch.rand([n, co], generator=g) + 0.5).to(DEV)
    rx = torch.randn([n, co, h, w], generator=g).to(dtype).to(DEV)
    other = torch.randn([n, co, h, w], generator=g).to(dtype).to(DEV)
    plan = cp.plan_conv(1, 1)
    have = cp.supports_addend(dy, plan, co, reduce=reduce, mask=masked)
    assert have == (not reduce or co in (32, 64))
    if not have:
        with pytest.raises(RuntimeError):
            cp.conv2d(dy, plan.pack(wt), plan, co, out_scale=s, gain=0.5, reduce_x=rx, addend=other)
        return
    kw = dict(mask_y=saved, mask_alpha=1.0, mask_clamp=2.0) if masked else {}
    kw.update(reduce_x=rx if reduce else None, out_scale=s, gain=0.5)
    plain = cp.conv2d(dy, plan.pack(wt), plan, co, **kw)
    fused = cp.conv2d(dy, plan.pack(wt), plan, co, addend=other, **kw)
    if reduce:
        (plain, r0), (fused, r1) = plain, fused
        close(r1, r0, 1e-6, 'reduction')
    assert torch.equal(fused, plain + other)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(3, 32, 3, 64, 64), (2, 3, 32, 64, 96), (2, 72, 3, 48, 96), (2, 4, 40, 64, 64), (3, 64, 3, 130, 136)],
                         ids=['x32_g3', 'x3_g32', 'x72_g3', 'x4_g40', 'x64_g3_big'])
@pytest.mark.parametrize('masked', [False, True])
def test_weight_gradient(shape, dtype, masked):
    from ga_gan_amd import conv2d_plugin as cp
    n, cx, cg, h, w = shape
    g = torch.Generator().manual_seed(cx * 7 + cg)
    x = torch.randn([n, cx, h, w], generator=g).to(dtype)
    dy = torch.randn([n, cg, h, w], generator=g).to(dtype)
    saved = (torch.randn([n, cg, h, w], generator=g) * 1.2).to(dtype)
    xs, gs = torch.rand([n, cx], generator=g) + 0.5, torch.rand([n, cg], generator=g) + 0.5
    kw = dict(mask_y=saved.to(DEV), mask_alpha=0.2, mask_clamp=2.0) if masked else {}
    if masked:
        assert cp.wgrad_supports_mask(x.to(DEV), dy.to(DEV), 1, 1)
    dw = cp.conv2d_wgrad(x.to(DEV), dy.to(DEV), 1, 1, x_scale=xs.to(DEV), g_scale=gs.to(DEV), **kw)
    gin = dy.float()
    if masked:
        m = torch.where(saved.float() > 0, 1.0, 0.2) * (saved.float().abs() < 2.0)
        gin = (gin * m).to(dtype).float()
    want = torch.einsum('nghw,nxhw,ng,nx->gx', gin.double(), x.double(), gs.double(), xs.double()).reshape(cg, cx, 1, 1)
    assert dw.shape == want.shape and dw.dtype == torch.float32
    close(dw, want, 1e-4)


def test_fused_to_rgb_layer_matches_composite():
    """The to-RGB layer through `fused_layer.conv_layer` (streaming kernels, masked backward) against the same layer
    from the differentiable primitive ops."""
    from ga_gan_amd.torch_utils.ops import fused_layer
    g = torch.Generator(device=DEV).manual_seed(3)
    n, ci, res = 2, 32, 128
    x = torch.randn(n, ci, res, res, device=DEV, generator=g).to(torch.bfloat16).requires_grad_()
    w = (torch.randn(3, ci, 1, 1, device=DEV, generator=g) / ci ** 0.5).to(torch.bfloat16).requires_grad_()
    s = (1 + 0.1 * torch.randn(n, ci, device=DEV, generator=g)).requires_grad_()
    b = torch.randn(3, device=DEV, generator=g).to(torch.bfloat16).requires_grad_()
    dy = torch.randn(n, 3, res, res, device=DEV, generator=g).to(torch.bfloat16)
    y = fused_layer.conv_layer(x, w, styles=s, bias=b, act='linear', clamp=1.0)
    got = torch.autograd.grad(y, [x, w, s, b], dy)
    pre = torch.einsum('nchw,oc,nc->nohw', x.double(), w.double()[:, :, 0, 0], s.double()) + b.double().reshape(1, 3, 1, 1)
    ref = pre.clamp(-1.0, 1.0)
    want = torch.autograd.grad(ref, [x, w, s, b], dy.double() * (y.double().abs() < 1.0))
    close(y, ref, 2 ** -6)
    for name, a, r, tol in zip('x w s b'.split(), got, want, (2 ** -6, 2e-3, 2e-3, 2e-2)):
        close(a, r, tol, name)
