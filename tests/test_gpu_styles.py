"""sg_styles / sg_styles_backward (ga_gan_amd/torch_utils/ops/styles.py) against the same expressions written with the
framework's operators in float64: values and every gradient, the strided latent row as the synthesis network passes
it, layers without demodulation, and the higher-order path (a gradient of a gradient, as the path-length
regulariser takes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ops():
    from ga_gan_amd.torch_utils.ops import styles
    return styles


def _case(n, k, c, o, seed, strided=True):
    g = torch.Generator().manual_seed(seed)
    ws = torch.randn(n, 3, k, generator=g)
    w = ws[:, 1] if strided else ws[:, 1].contiguous()
    a = torch.randn(c, k, generator=g)
    b = torch.randn(c, generator=g) + 1
    wsq_t = torch.rand(c, o, generator=g) * 9 if o else None
    return w, a, b, wsq_t


def _rel(x, y):
    x, y = x.detach().double().cpu(), y.detach().double().cpu()
    return float((x - y).norm() / max(float(y.norm()), 1e-30))


@pytest.mark.parametrize('n,k,c,o', [(32, 512, 512, 512), (16, 512, 64, 32), (3, 512, 100, 8), (8, 512, 256, 0), (1, 64, 5, 4), (9, 36, 17, 20),
                                     (20, 1024, 40, 300)])
def test_styles_and_gradients_against_float64(n, k, c, o):
    ops = _ops()
    gains = (1 / k ** 0.5, 1.0, 0.7)
    w, a, b, wsq_t = _case(n, k, c, o, seed=n + c)
    leaves64 = [t.double().requires_grad_(True) if t is not None else None for t in (w, a, b, wsq_t)]
    s64, d64 = ops.reference(*leaves64, *gains)
    leaves = [t.to(DEV).requires_grad_(True) if t is not None else None for t in (w, a, b, wsq_t)]
    view = leaves[0]
    if n > 1:    # a row of ws as it lies: stride 3 k between samples
        base = torch.zeros(n, 3, k, device=DEV)
        base[:, 1] = w.to(DEV)
        base.requires_grad_(True)
        view = base[:, 1]
    s, d = ops.styles(view, leaves[1], leaves[2], leaves[3], *gains)
    assert s.shape == (n, c) and (d is None) == (o == 0)
    assert _rel(s, s64) < 2e-6
    gs = torch.randn(n, c, generator=torch.Generator().manual_seed(1))
    outs64, outs, seeds = [s64], [s], [gs]
    if o:
        assert _rel(d, d64) < 2e-6
        outs64.append(d64); outs.append(d)
        seeds.append(torch.randn(n, o, generator=torch.Generator().manual_seed(2)))
    want = torch.autograd.grad(outs64, [t for t in leaves64 if t is not None], [g.double() for g in seeds])
    ins = [base if n > 1 else leaves[0]] + [t for t in leaves[1:] if t is not None]
    got = torch.autograd.grad(outs, ins, [g.to(DEV) for g in seeds])
    got = [got[0][:, 1] if n > 1 else got[0]] + list(got[1:])
    for name, g_, w_ in zip(['w', 'a', 'b', 'wsq_t'], got, want):
        assert _rel(g_, w_) < 1e-5, name


def test_styles_only_one_output_used():
    """Gradient through the coefficients alone and through the styles alone (the other arrives as None)."""
    ops = _ops()
    w, a, b, wsq_t = _case(4, 512, 128, 64, seed=5, strided=False)
    for pick in (0, 1):
        leaves64 = [t.double().requires_grad_(True) for t in (w, a, b, wsq_t)]
        leaves = [t.to(DEV).requires_grad_(True) for t in (w, a, b, wsq_t)]
        want = torch.autograd.grad(ops.reference(*leaves64, 0.05, 1.0, 1.0)[pick].square().sum(), leaves64, allow_unused=True)
        got = torch.autograd.grad(ops.styles(*leaves, 0.05, 1.0, 1.0)[pick].square().sum(), leaves, allow_unused=True)
        for name, g_, w_ in zip(['w', 'a', 'b', 'wsq_t'], got, want):
            if w_ is None:
                assert g_ is None or float(g_.abs().max()) == 0, name
            else:
                assert _rel(g_, w_) < 1e-5, (pick, name)


def test_styles_gradient_of_gradient():
    """create_graph=True differentiates the operators' version: second-order terms against float64."""
    ops = _ops()
    w, a, b, wsq_t = _case(4, 512, 96, 48, seed=9, strided=False)

    def second(fn, leaves):
        s, d = fn(*leaves, 0.04, 1.0, 1.0)
        g, = torch.autograd.grad((s.sin() * 0.3).sum() + d.sum(), leaves[0], create_graph=True)
        return torch.autograd.grad(g.square().sum(), [leaves[0], leaves[1]])

    want = second(ops.reference, [t.double().requires_grad_(True) for t in (w, a, b, wsq_t)])
    got = second(ops.styles, [t.to(DEV).requires_grad_(True) for t in (w, a, b, wsq_t)])
    for g_, w_ in zip(got, want):
        assert _rel(g_, w_) < 1e-4
