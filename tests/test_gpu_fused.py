"""GPU parity of the one-kernel convolution layers against the same layer composed from the
primitive ops (which the golden-vector tests pin to the reference), values and all gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).norm() / b.norm().clamp_min(1e-30)).detach())


def _inputs(n, ci, co, res, k, dtype, modulated, noise, seed):
    g = torch.Generator().manual_seed(seed)
    dev = 'cuda'
    x = torch.randn(n, ci, res, res, generator=g).to(dev, dtype).requires_grad_()
    w = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).to(dev, dtype).requires_grad_()
    styles = (1 + 0.3 * torch.randn(n, ci, generator=g)).to(dev).requires_grad_() if modulated else None
    dcoefs = (1 + 0.3 * torch.rand(n, co, generator=g)).to(dev).requires_grad_() if modulated else None
    bias = (0.1 * torch.randn(co, generator=g)).to(dev, dtype).requires_grad_()
    strength = torch.tensor(0.3, device=dev).requires_grad_() if noise else None
    return x, w, styles, dcoefs, bias, strength


CASES = [
    # n, ci, co, res, k, up, down, modulated, noise, act, clamp
    (2, 24, 40, 16, 3, 1, 1, True, True, 'lrelu', 256.0),      # several images per pixel tile
    (3, 16, 8, 64, 3, 1, 1, True, True, 'lrelu', 256.0),       # one image per tile, several tiles per image
    (2, 32, 16, 32, 3, 2, 1, True, True, 'lrelu', 256.0),      # up-sampling layer
    (2, 32, 3, 32, 1, 1, 1, True, False, 'linear', 256.0),     # to-RGB
    (2, 16, 24, 32, 3, 1, 2, False, False, 'lrelu', 256.0),    # discriminator down-sampling conv
    (2, 16, 24, 32, 1, 1, 2, False, False, 'linear', None),    # discriminator skip
    (2, 3, 16, 32, 1, 1, 1, False, False, 'lrelu', 0.5),       # from-RGB, clamp active
    (5, 20, 72, 8, 3, 1, 1, True, True, 'lrelu', None),        # ragged channel counts
    (2, 96, 80, 32, 3, 1, 1, True, True, 'lrelu', 256.0),      # wide layer: the 3x3 unit-stride kernels, forward and both gradients
    (2, 72, 136, 40, 3, 1, 1, True, False, 'linear', None),
    # up-sampling layers of the one-pass kernel (transposed convolution + filter + epilogue):
    (2, 64, 32, 64, 3, 2, 1, True, True, 'lrelu', 256.0),      # whole composed kernel resident in LDS, two channel chunks
    (1, 32, 16, 40, 3, 2, 1, True, True, 'lrelu', 256.0),      # one chunk, ragged width, half a channel group
    (2, 96, 48, 72, 3, 2, 1, True, False, 'linear', None),     # streamed weights, two channel groups, ragged tiles
    (1, 128, 64, 32, 3, 2, 1, False, True, 'lrelu', 256.0),    # narrow rows, four chunks, no modulation
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', CASES, ids=[f'c{i}' for i in range(len(CASES))])
def test_fused_layer_matches_composite(case, dtype):
    from ga_gan_amd.torch_utils.ops import fused_layer, upfirdn2d
    n, ci, co, res, k, up, down, modulated, noise, act, clamp = case
    x, w, styles, dcoefs, bias, strength = _inputs(n, ci, co, res, k, dtype, modulated, noise, seed=res * 31 + ci)
    assert fused_layer.supported(x, w, up, down) or dtype == torch.float32
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device='cuda')
    out_res = res * up // down
    nz = torch.randn(out_res, out_res, device='cuda', generator=torch.Generator('cuda').manual_seed(3)) if noise else None
    pad = k // 2
    y = fused_layer.conv_layer(x, w, f=f, up=up, down=down, padding=pad, flip_weight=(up == 1), styles=styles, dcoefs=dcoefs,
                               noise=nz, noise_strength=strength, bias=bias, act=act, clamp=clamp)

    from ga_gan_amd.torch_utils.ops import bias_act, conv2d_resample

    def composite(cdtype):
        """The same layer from the primitive ops on independent leaves holding the same (already rounded) values."""
        leaves = [t.detach().clone().requires_grad_() if t is not None else None for t in (x, w, styles, dcoefs, bias, strength)]
        x2, w2, s2, d2, b2, g2 = leaves
        x2c, w2c, b2c = x2.to(cdtype), w2.to(cdtype), b2.to(cdtype)
        xm = x2c * s2.to(cdtype).reshape(n, ci, 1, 1) if modulated else x2c
        c = conv2d_resample.conv2d_resample(xm, w2c, f=f, up=up, down=down, padding=pad, flip_weight=(up == 1))
        if modulated:
            c = c * d2.to(cdtype).reshape(n, co, 1, 1)
        if noise:
            c = c + (nz * g2).to(cdtype)
        out = bias_act.bias_act(c, b2c, act=act, clamp=clamp)
        out.backward(dy.to(cdtype))
        return out.detach(), [t.grad if t is not None else None for t in leaves]

    dy = torch.randn(y.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(5)).to(dtype)
    y.backward(dy)
    grads = [t.grad if t is not None else None for t in (x, w, styles, dcoefs, bias, strength)]
    exact_y, exact_grads = composite(torch.float32)
    names = ('x', 'w', 'styles', 'dcoefs', 'bias', 'noise_strength')
    assert y.shape == exact_y.shape and y.dtype == dtype
    if dtype == torch.float32:
        assert _rel(y, exact_y) < 2e-5
        for name, a, b in zip(names, grads, exact_grads):
            if b is not None:
                assert _rel(a, b) < 5e-5, name
        return
    # bfloat16: judged against exact arithmetic on the same inputs, and allowed the error the layer composed from
    # the primitive ops in bfloat16 makes (rounding of intermediates flips a few leaky-ReLU slopes and clamp masks,
    # which dominates the gradient error of either formulation)
    lowp_y, lowp_grads = composite(dtype)
    assert _rel(y, exact_y) < max(1.5 * _rel(lowp_y, exact_y), 4e-3)
    for name, a, b, c in zip(names, grads, exact_grads, lowp_grads):
        if b is None:
            continue
        assert a is not None, name
        if name == 'noise_strength':
            # one scalar, a sum of ~N signed terms that nearly cancel: a handful of flipped slopes moves it by
            # several percent, so the error is measured against the sum of the terms' magnitudes instead
            scale = float((dy.float().abs() * nz.abs()).sum())
            assert abs(float(a) - float(b)) < 2e-3 * scale, (float(a), float(b), scale)
            continue
        assert _rel(a, b) < max(1.5 * _rel(c, b), 5e-3), (name, _rel(a, b), _rel(c, b))


def test_fused_layer_double_backward_matches_composite():
    """R1 / path-length style second-order use: gradient of a gradient through the fused layer."""
    from ga_gan_amd.torch_utils.ops import fused_layer, upfirdn2d, bias_act, conv2d_resample
    n, ci, co, res, k = 2, 8, 12, 16, 3
    x, w, styles, dcoefs, bias, strength = _inputs(n, ci, co, res, k, torch.float32, True, True, seed=11)
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device='cuda')
    nz = torch.randn(res, res, device='cuda', generator=torch.Generator('cuda').manual_seed(3))

    def second_order(fused):
        leaves = [t.detach().clone().requires_grad_() for t in (x, w, styles, dcoefs, bias, strength)]
        x2, w2, s2, d2, b2, g2 = leaves
        if fused:
            y = fused_layer.conv_layer(x2, w2, f=f, padding=1, styles=s2, dcoefs=d2, noise=nz, noise_strength=g2, bias=b2,
                                       act='lrelu', clamp=256.0)
        else:
            c = conv2d_resample.conv2d_resample(x2 * s2.reshape(n, ci, 1, 1), w2, f=f, padding=1)
            y = bias_act.bias_act(c * d2.reshape(n, co, 1, 1) + nz * g2, b2, act='lrelu', clamp=256.0)
        gx, gs = torch.autograd.grad(y.square().sum(), [x2, s2], create_graph=True)
        (gx.square().sum() + gs.square().sum()).backward()
        return [t.grad for t in leaves]

    for name, a, b in zip(('x', 'w', 'styles', 'dcoefs', 'bias', 'noise_strength'), second_order(True), second_order(False)):
        if b is None:
            assert a is None or float(a.abs().max()) == 0, name
            continue
        assert _rel(a, b) < 1e-4, name


def test_conv2d_style_reduction_rows():
    """The in-kernel reduction: partial rows (one image per tile) and atomics (several images per tile)."""
    from ga_gan_amd import conv2d_plugin as cp
    for n, c, res in ((2, 16, 64), (6, 16, 8), (3, 80, 32)):
        g = torch.Generator().manual_seed(res)
        x = torch.randn(n, c, res, res, generator=g).cuda()
        w = (torch.randn(c, c, 3, 3, generator=g) / 12).cuda()
        r = torch.randn(n, c, res, res, generator=g).cuda()
        plan = cp.plan_conv(3, 3, stride=1, padding=1)
        y, red = cp.conv2d(x, plan.pack(w), plan, c, reduce_x=r)
        ref = torch.nn.functional.conv2d(x.cpu().double(), w.cpu().double(), padding=1)
        assert _rel(y.cpu(), ref) < 1e-5
        assert _rel(red.cpu(), (ref * r.cpu().double()).sum(dim=[2, 3])) < 1e-5


F64_CASES = [
    # n, ci, co, res, k, up, magnitude of x, magnitude of dy
    (1, 512, 512, 16, 3, 1, 1.0, 1.0),
    (1, 512, 512, 16, 3, 2, 1.0, 1.0),
    (1, 512, 3, 32, 1, 1, 1.0, 1.0),
    (4, 256, 256, 64, 3, 1, 1.0, 1.0),
    (2, 128, 128, 32, 3, 1, 3e4, 1e-9),      # activations near the top of a half's range, gradients far below it
    (2, 64, 96, 32, 3, 1, 1e-7, 1e5),
    (2, 36, 40, 12, 3, 1, 1.0, 1.0),         # ragged: the scalar staging path
]


@pytest.mark.parametrize('case', F64_CASES, ids=[f'f{i}' for i in range(len(F64_CASES))])
def test_float32_layers_against_float64(case):
    """The float32 kernels multiply in half precision pieces after scaling each operand tensor into a half's range:
    values and every gradient of wide low-resolution layers stay within 5e-6 relative L2 of the same layer in
    float64 arithmetic (a float32 evaluation lies at 1e-6), whatever the magnitude of activations and gradients."""
    import torch.nn.functional as F
    from ga_gan_amd.torch_utils.ops import fused_layer, upfirdn2d
    n, ci, co, res, k, up, mag_x, mag_dy = case
    g = torch.Generator().manual_seed(res + ci)
    x = torch.randn(n, ci, res, res, generator=g) * mag_x
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    s = 1 + 0.3 * torch.randn(n, ci, generator=g)
    d = 1 + 0.3 * torch.rand(n, co, generator=g)
    b = 0.1 * torch.randn(co, generator=g) * mag_x
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device='cuda')
    leaves = [t.cuda().requires_grad_() for t in (x, w, s, d, b)]
    y = fused_layer.conv_layer(leaves[0], leaves[1], f=f, up=up, down=1, padding=k // 2, flip_weight=(up == 1), styles=leaves[2],
                               dcoefs=leaves[3], bias=leaves[4], act='lrelu' if k == 3 else 'linear', clamp=None)
    dy = torch.randn(y.shape, generator=g) * mag_dy
    y.backward(dy.cuda())
    exact = [t.double().requires_grad_() for t in (x, w, s, d, b)]
    xm = exact[0] * exact[2].reshape(n, ci, 1, 1)
    if up == 1:
        c = F.conv2d(xm, exact[1], padding=k // 2)
    else:
        c = F.conv_transpose2d(xm, exact[1].transpose(0, 1), stride=2)
        f1 = torch.tensor([1., 3, 3, 1], dtype=torch.float64)
        f2 = torch.outer(f1, f1)
        f2 = f2 / f2.sum() * 4
        c = F.conv2d(F.pad(c, [1, 1, 1, 1]).reshape(n * co, 1, c.shape[2] + 2, c.shape[3] + 2), f2[None, None]).reshape(n, co, 2 * res, 2 * res)
    c = c * exact[3].reshape(n, co, 1, 1) + exact[4].reshape(1, co, 1, 1)
    if k == 3:
        c = F.leaky_relu(c, 0.2) * 2 ** 0.5
    c.backward(dy.double())
    assert _rel(y.detach().cpu(), c.detach()) < 5e-6
    for name, a, e in zip('x w styles dcoefs bias'.split(), leaves, exact):
        assert _rel(a.grad.cpu(), e.grad) < 5e-6, (name, _rel(a.grad.cpu(), e.grad))


@pytest.mark.parametrize('trainable_styles', [True, False], ids=['styles', 'frozen'])
@pytest.mark.parametrize('co_in', [32, 64, 48], ids=['c32', 'c64', 'c48'])
def test_to_rgb_layer_that_passes_its_input_on(co_in, trainable_styles):
    """`conv_layer(..., also_input=True)` returns (x, y); a second consumer fed from the returned x gives x the same
    gradient - bit for bit in bfloat16 - as the plain layer whose input autograd sums over both consumers, whether the
    layer's data-gradient kernel takes the other gradient as an addend (32 / 64 channels with the style reduction, any
    multiple of four without) or the sum stays a step of its own."""
    from ga_gan_amd.torch_utils.ops import fused_layer
    n, res = 3, 64
    results = []
    for branch in (True, False):
        x, w, styles, _, bias, _ = _inputs(n, co_in, 3, res, 1, torch.bfloat16, True, False, seed=co_in)
        styles.requires_grad_(trainable_styles)
        other = torch.randn(x.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(9)).to(torch.bfloat16)
        kw = dict(styles=styles, bias=bias, act='linear', clamp=256.0)
        if branch:
            x2, y = fused_layer.conv_layer(x, w, also_input=True, **kw)
            assert x2.data_ptr() == x.data_ptr()
        else:
            x2, y = x, fused_layer.conv_layer(x, w, **kw)
        dy = torch.randn(y.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(5)).to(torch.bfloat16)
        ((y * dy).sum() + (x2 * other).sum()).backward()
        results.append((y.detach(), x.grad, w.grad, styles.grad, bias.grad))
    for name, a, b in zip(('y', 'dx', 'dw', 'dstyles', 'dbias'), *results):
        if b is None:
            assert a is None, name
        elif name in ('y', 'dx'):
            assert torch.equal(a, b), name
        else:
            assert _rel(a, b) < 1e-5, name      # (float atomics: the order of the partial sums differs from run to run)


def test_branch_layer_double_backward():
    """The passed-through input under `create_graph` (the path-length regulariser differentiates a gradient that reached
    the generator through both consumers of a block's output)."""
    from ga_gan_amd.torch_utils.ops import fused_layer
    n, ci, res = 2, 16, 32

    def second_order(branch):
        x, w, styles, _, bias, _ = _inputs(n, ci, 3, res, 1, torch.float32, True, False, seed=4)
        kw = dict(styles=styles, bias=bias, act='linear', clamp=256.0)
        if branch:
            x2, y = fused_layer.conv_layer(x, w, also_input=True, **kw)
        else:
            x2, y = x, fused_layer.conv_layer(x, w, **kw)
        gx, gs = torch.autograd.grad(y.square().sum() + x2.pow(3).sum(), [x, styles], create_graph=True)
        (gx.square().sum() + gs.square().sum()).backward()
        return x.grad, w.grad, styles.grad, bias.grad

    for name, a, b in zip(('x', 'w', 'styles', 'bias'), second_order(True), second_order(False)):
        assert _rel(a, b) < 1e-5, name
