"""Per-plane scale / dot product (`sg_scale_planes`, `sg_dot_planes`) and the autograd pair built on them, against the
broadcast multiply and multiply-and-sum they stand for (reference training/networks.py:642, :648-650), through second
order."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = {torch.float32: 1e-5, torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}


def close(a, b, tol, what):
    a, b = a.detach().double(), b.detach().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-6), what


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(3, 5, 8, 8), (2, 7, 40, 72), (1, 3, 256, 200), (4, 16, 4, 4)], ids=str)
def test_scale_and_dot(shape, dtype):
    from ga_gan_amd.torch_utils.ops import planes
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(shape, device=DEV, generator=g).to(dtype)
    y = torch.randn(shape, device=DEV, generator=g).to(dtype)
    s = torch.randn(shape[:2], device=DEV, generator=g)
    got = planes.scale_planes(x, s)
    assert got.dtype == dtype
    close(got, x.double() * s.double().reshape(*shape[:2], 1, 1), TOL[dtype], 'scale')
    d = planes.dot_planes(x, y)
    assert d.dtype == torch.float32 and tuple(d.shape) == shape[:2]
    close(d, (x.double() * y.double()).sum([2, 3]), 1e-5, 'dot')


def test_gradients_through_second_order():
    from ga_gan_amd.torch_utils.ops import planes
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn([2, 6, 16, 24], device=DEV, generator=g, dtype=torch.float64).float().requires_grad_()
    s = torch.randn([2, 6], device=DEV, generator=g).requires_grad_()
    w = torch.randn([2, 6, 16, 24], device=DEV, generator=g)

    def first(fn):
        y = fn(x, s)
        gx, gs = torch.autograd.grad((y * w).sum() + y.square().sum(), [x, s], create_graph=True)
        hx, hs = torch.autograd.grad(gx.square().sum() + gs.square().sum(), [x, s])
        return y, gx, gs, hx, hs

    got = first(planes.scale_planes)
    want = first(lambda a, b: a * b.reshape(2, 6, 1, 1))
    for name, a, b in zip(('y', 'gx', 'gs', 'hx', 'hs'), got, want):
        close(a, b, 1e-4, name)


def test_odd_planes_take_the_plain_path():
    from ga_gan_amd.torch_utils.ops import planes
    x = torch.randn([2, 3, 5, 7], device=DEV).to(torch.bfloat16)
    s = torch.randn([2, 3], device=DEV)
    close(planes.scale_planes(x, s), x.double() * s.double().reshape(2, 3, 1, 1), 2 ** -6, 'scale')
    close(planes.dot_planes(x, x), (x.double() * x.double()).sum([2, 3]), 2 ** -6, 'dot')
