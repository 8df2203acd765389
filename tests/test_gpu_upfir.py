"""GPU parity of `sg_conv_up2_fir` (an up-sampling layer in one pass) against the two-pass form it replaces
(`sg_conv2d` transposed convolution, then `sg_upfirdn2d_epilogue`) and against float64 arithmetic on the same inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


CASES = [
    # n, ci, co, h, w, dtype, per-sample noise
    (2, 64, 32, 32, 64, torch.bfloat16, False),
    (3, 64, 32, 9, 96, torch.float16, True),      # ragged rows (h not a multiple of the tile), 1.5 tiles wide
    (2, 160, 80, 16, 32, torch.bfloat16, True),   # five chunks, three channel groups (the last one half empty)
    (1, 32, 8, 2, 32, torch.bfloat16, False),     # smallest geometry
]


@pytest.mark.parametrize('case', CASES, ids=[f'u{i}' for i in range(len(CASES))])
def test_conv_up2_fir_matches_two_pass_and_float64(case):
    from ga_gan_amd import conv2d_plugin as conv, layer_plugin as layer
    from ga_gan_amd.torch_utils.ops import upfirdn2d
    n, ci, co, h, w, dtype, per_sample = case
    g = torch.Generator().manual_seed(ci * 7 + h)
    x = torch.randn(n, ci, h, w, generator=g).to('cuda', dtype)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).to('cuda', dtype)
    styles = (1 + 0.3 * torch.randn(n, ci, generator=g)).cuda()
    dcoefs = (1 + 0.3 * torch.rand(n, co, generator=g)).cuda()
    bias = (0.1 * torch.randn(co, generator=g)).to('cuda', dtype)
    noise = torch.randn(*((n, 1) if per_sample else ()), 2 * h, 2 * w, generator=g).cuda()
    strength = torch.tensor(0.3, device='cuda')
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device='cuda')
    ep = dict(scale=dcoefs, noise=noise, noise_gain=strength, bias=bias, act='lrelu', alpha=0.2, gain=2 ** 0.5, clamp=256.0)

    assert layer.conv_up2_fir_supported(x, co, 'lrelu', 0.2, 2 ** 0.5)
    m, fy4 = layer.separable_taps(f)
    y = layer.conv_up2_fir(x, layer.compose_up2_weights(wt, m), co, fy4, in_scale=styles, **ep)
    assert y.shape == (n, co, 2 * h, 2 * w) and y.dtype == dtype

    plan = conv.plan_conv_transpose(3, 3, stride=2, padding=0)
    extra = (-(2 * w + 1)) % 8
    t = conv.conv2d(x, plan.pack(wt), plan, co, in_scale=styles, out_hw=(2 * h + 1, 2 * w + 1 + extra))
    two_pass = layer.blur_epilogue(t, f, (1, 1 - extra, 1, 1), False, 4.0, **ep)

    xd = x.double() * styles.double().reshape(n, ci, 1, 1)
    td = torch.nn.functional.conv_transpose2d(xd, wt.double().transpose(0, 1), stride=2)
    fd = f.double().flip([0, 1])[None, None].repeat(co, 1, 1, 1)
    ud = torch.nn.functional.conv2d(torch.nn.functional.pad(td, [1, 1, 1, 1]), fd, groups=co) * 4
    zd = ud * dcoefs.double().reshape(n, co, 1, 1) + noise.double().reshape(-1, 1, 2 * h, 2 * w) * 0.3 + bias.double().reshape(1, co, 1, 1)
    exact = (torch.nn.functional.leaky_relu(zd, 0.2) * 2 ** 0.5).clamp(-256, 256)

    err, err_two_pass = _rel(y, exact), _rel(two_pass, exact)
    # no intermediate is rounded to 16 bits here, so the one-pass result is at least as close to exact arithmetic
    # as the two-pass one; both carry the rounding of the output (2^-9 relative for bfloat16, 2^-12 for float16)
    bound = 4e-3 if dtype == torch.bfloat16 else 6e-4
    assert err < bound, (err, err_two_pass)
    assert err < 1.2 * err_two_pass + 1e-4, (err, err_two_pass)


def test_conv_up2_fir_rejects_what_it_does_not_cover():
    from ga_gan_amd import layer_plugin as layer
    x = torch.randn(1, 32, 8, 24, device='cuda').to(torch.bfloat16)       # rows of 24 pixels: narrower than a strip pair
    assert not layer.conv_up2_fir_supported(x, 16)
    assert not layer.conv_up2_fir_supported(x.float(), 16)
    x = torch.randn(1, 24, 8, 64, device='cuda').to(torch.bfloat16)       # fewer than 32 input channels
    assert not layer.conv_up2_fir_supported(x, 16)
    f = torch.rand(4, 4, device='cuda')                                    # not separable
    assert layer.separable_taps(f) is None
