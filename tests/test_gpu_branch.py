"""Filter + addend in one pass (`sg_upfirdn2d_add`) and the branch op built on it: the gradient of a tensor that feeds a
down-sampling filter and a second consumer (the input of a residual discriminator block, reference
training/networks.py:792-808).  The fused pass performs the arithmetic of the two separate steps - filter, round, add,
round - so results are compared for equality."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _filter():
    from ga_gan_amd.torch_utils.ops import upfirdn2d
    return upfirdn2d.setup_filter([1, 3, 3, 1], device=DEV)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 6, 32, 32), (3, 5, 40, 72), (1, 3, 520, 520), (2, 1, 64, 64)], ids=str)
def test_upfirdn2d_add_equals_filter_then_add(shape, dtype):
    from ga_gan_amd import upfirdn2d_plugin as up
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    n, c, h, w = shape
    x = torch.randn(shape, device=DEV, generator=g).to(dtype)
    a = torch.randn([n, c, 2 * h, 2 * w], device=DEV, generator=g).to(dtype)
    f = _filter()
    args = (2, 2, 1, 1, 2, 1, 2, 1, True, 4.0)         # the adjoint of downsample2d with a 4-tap filter
    want = up.upfirdn2d(x, f, *args) + a
    got = up.upfirdn2d_add(x, f, *args, addend=a)
    assert got is not None and got.dtype == dtype
    assert torch.equal(got, want)


def test_upfirdn2d_add_declines_other_geometries():
    from ga_gan_amd import upfirdn2d_plugin as up
    f = _filter()
    x = torch.randn([2, 4, 32, 32], device=DEV)
    assert up.upfirdn2d_add(x, f, 2, 2, 1, 1, 2, 1, 2, 1, True, 4.0, addend=torch.zeros([2, 4, 64, 64], device=DEV)) is None   # float32
    xb = x.to(torch.bfloat16)
    assert up.upfirdn2d_add(xb, f, 1, 1, 1, 1, 2, 1, 2, 1, True, 1.0, addend=torch.zeros([2, 4, 32, 32], device=DEV, dtype=torch.bfloat16)) is None   # no up-sampling
    assert up.upfirdn2d_add(xb, f, 2, 2, 1, 1, 2, 1, 2, 1, True, 4.0, addend=torch.zeros([2, 4, 64, 60], device=DEV, dtype=torch.bfloat16)) is None   # wrong shape


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_branch_gradient_equals_separate_sum(dtype):
    from ga_gan_amd.torch_utils.ops import upfirdn2d
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn([2, 8, 64, 64], device=DEV, generator=g).to(dtype).requires_grad_()
    wa = torch.randn([2, 8, 64, 64], device=DEV, generator=g).to(dtype)
    wb = torch.randn([2, 8, 32, 32], device=DEV, generator=g).to(dtype)
    f = _filter()
    pad = [1, 1, 1, 1]
    x1, xd = upfirdn2d.branch(x, f, down=2, padding=pad)
    assert torch.equal(xd, upfirdn2d.upfirdn2d(x, f, down=2, padding=pad))
    got, = torch.autograd.grad((x1 * wa).sum() + (xd * wb).sum(), [x])
    want, = torch.autograd.grad((x * wa).sum() + (upfirdn2d.upfirdn2d(x, f, down=2, padding=pad) * wb).sum(), [x])
    assert torch.equal(got, want)
    # one consumer only
    x1, xd = upfirdn2d.branch(x, f, down=2, padding=pad)
    got, = torch.autograd.grad((xd * wb).sum(), [x])
    want, = torch.autograd.grad((upfirdn2d.upfirdn2d(x, f, down=2, padding=pad) * wb).sum(), [x])
    assert torch.equal(got, want)
    # second order: the branch stays differentiable
    x1, xd = upfirdn2d.branch(x, f, down=2, padding=pad)
    g1, = torch.autograd.grad((x1 * wa).sum() + (xd.square() * wb).sum(), [x], create_graph=True)
    g2, = torch.autograd.grad(g1.square().sum(), [x])
    y = upfirdn2d.upfirdn2d(x, f, down=2, padding=pad)
    h1, = torch.autograd.grad((x * wa).sum() + (y.square() * wb).sum(), [x], create_graph=True)
    h2, = torch.autograd.grad(h1.square().sum(), [x])
    assert torch.allclose(g2.float(), h2.float(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-5, atol=1e-3)


def test_discriminator_block_with_and_without_branch():
    """A residual block in bfloat16: output and input gradient with the branch op and the sum formed by the skip
    layer's kernel equal those of the plain form (filter, layers, then the sum in a pass of its own)."""
    from ga_gan_amd.training import networks
    torch.manual_seed(0)
    block = networks.DiscriminatorBlock(32, 32, 64, resolution=128, first_layer_idx=0, use_fp16=True, conv_clamp=256,
                                        img_channels=3, architecture='resnet').to(DEV).requires_grad_(True)
    x = torch.randn([4, 32, 128, 128], device=DEV).requires_grad_()
    dy = torch.randn([4, 64, 64, 64], device=DEV).to(networks.LOW_PRECISION_DTYPE)
    results = []
    for use_branch in (True, False):
        saved = networks.Conv2dLayer.forward
        if not use_branch:
            def plain(self, x, gain=1, also_input=False, residual=None, _f=saved):
                y = _f(self, x, gain=gain)
                if residual is not None:      # (the sum in a pass of its own, after both branches)
                    return y.add_(residual(x))
                return (x, y) if also_input else y
            networks.Conv2dLayer.forward = plain
        try:
            y, _ = block(x, None)
            grads = torch.autograd.grad(y, [x] + list(block.parameters()), dy)
        finally:
            networks.Conv2dLayer.forward = saved
        results.append((y, grads))
    (ya, ga), (yb, gb) = results
    assert torch.equal(ya, yb)
    assert torch.equal(ga[0], gb[0])                       # the block input's gradient: the same arithmetic in one pass
    # Weight gradients are summed with float atomics, in an order that varies from launch to launch, and those of the
    # 16-bit layers are then rounded to bfloat16 (the type the weight was cast to): a sum that differs in its last
    # float32 bits can land on the other side of a rounding boundary, one bfloat16 ulp (2^-7 of the largest element).
    for a, b in zip(ga[1:], gb[1:]):
        assert torch.allclose(a.float(), b.float(), rtol=2.0 ** -7, atol=2.0 ** -7 * float(b.abs().max()))
