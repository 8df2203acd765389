"""The CPU oracle against the reference's own outputs (tests/golden/*.npz)."""
import numpy as np
import pytest
import torch

import golden_io
from oracle import pointwise, resample, stylegan2

TOL = 2e-6  # fp32 round-off between two orderings of the same arithmetic


def close(a, b, tol=TOL):
    scale = max(float(b.detach().abs().max()), 1.0)
    assert golden_io.max_err(a, b) <= tol * scale, (golden_io.max_err(a, b), scale)


@pytest.mark.parametrize('case', golden_io.load('upfirdn2d'), ids=lambda c: str(len(str(c.meta))))
def test_upfirdn2d(case):
    m = case.meta
    x = case.t('x', grad=True)
    f = case.get('f')
    dy = case.t('dy', grad=True)
    y = resample.upfirdn2d(x, f, up=m['up'], down=m['down'], padding=m['padding'], flip_filter=m['flip_filter'], gain=m['gain'])
    close(y, case.t('y'))
    gx, = torch.autograd.grad((y * dy).sum(), x, create_graph=True)
    close(gx, case.t('gx'))
    ggdy, = torch.autograd.grad((gx * case.t('v')).sum(), dy)
    close(ggdy, case.t('ggdy'))


def test_upfirdn2d_numpy_loops_agree():
    cases = golden_io.load('upfirdn2d')
    picked = [c for c in cases if c.np('x').size <= 2 * 8 * 9 * 11][::4]
    assert len(picked) >= 8
    for case in picked:
        m = case.meta
        y = resample.upfirdn2d_numpy(case.np('x'), case.np('f') if 'f' in case else None, up=m['up'], down=m['down'],
                                     padding=m['padding'], flip_filter=m['flip_filter'], gain=m['gain'])
        assert y.shape == case.np('y').shape
        np.testing.assert_allclose(y, case.np('y'), atol=5e-6 * max(1.0, np.abs(y).max()))


@pytest.mark.parametrize('case', golden_io.load('bias_act'), ids=lambda c: c.meta['act'])
def test_bias_act(case):
    m = case.meta
    x = case.t('x', grad=True)
    b = case.get('b', grad=True)
    dy = case.t('dy', grad=True)
    y = pointwise.bias_act(x, b, dim=m['dim'], act=m['act'], alpha=m['alpha'], gain=m['gain'], clamp=m['clamp'])
    close(y, case.t('y'))
    grads = torch.autograd.grad((y * dy).sum(), [x] + ([b] if b is not None else []), create_graph=True)
    close(grads[0], case.t('gx'))
    if b is not None:
        close(grads[1], case.t('gb'), 1e-5)
    second = torch.autograd.grad((grads[0] * case.t('v')).sum(), [dy, x], allow_unused=True)
    close(second[0], case.t('ggdy'))
    close(second[1] if second[1] is not None else torch.zeros_like(x), case.t('ggx'), 1e-5)


@pytest.mark.parametrize('case', golden_io.load('bias_act'), ids=lambda c: c.meta['act'])
def test_bias_act_kernel_modes_numpy(case):
    """The three `grad` modes of the native entry point, stated element-wise in float64."""
    m = case.meta
    idx, def_alpha, def_gain = pointwise.ACTIVATIONS[m['act']]
    alpha = def_alpha if m['alpha'] is None else m['alpha']
    gain = def_gain if m['gain'] is None else m['gain']
    clamp = -1 if m['clamp'] is None else m['clamp']
    x = case.np('x')
    step = int(np.prod(x.shape[m['dim'] + 1:]))
    b = case.np('b') if 'b' in case else None
    y = pointwise.bias_act_numpy(x, b, None, None, None, 0, step, idx, alpha, gain, clamp).reshape(x.shape)
    np.testing.assert_allclose(y, case.np('y'), atol=3e-6 * max(1, np.abs(y).max()))
    gx = pointwise.bias_act_numpy(case.np('dy'), b, x, case.np('y'), None, 1, step, idx, alpha, gain, clamp).reshape(x.shape)
    np.testing.assert_allclose(gx, case.np('gx'), atol=2e-5 * max(1, np.abs(gx).max()))
    if m['act'] not in ('linear', 'relu', 'lrelu'):
        ggx = pointwise.bias_act_numpy(case.np('v'), b, x, case.np('y'), case.np('dy'), 2, step, idx, alpha, gain, clamp)
        np.testing.assert_allclose(ggx.reshape(x.shape), case.np('ggx'), atol=1e-4 * max(1, np.abs(ggx).max()))


@pytest.mark.parametrize('case', golden_io.load('conv2d_resample'), ids=lambda c: f"u{c.meta['up']}d{c.meta['down']}g{c.meta['groups']}")
def test_conv2d_resample(case):
    m = case.meta
    x = case.t('x', grad=True)
    w = case.t('w', grad=True)
    y = resample.conv2d_resample(x, w, f=case.get('f'), up=m['up'], down=m['down'], padding=m['padding'],
                                 groups=m['groups'], flip_weight=m['flip_weight'])
    close(y, case.t('y'), 1e-5)
    gx, gw = torch.autograd.grad((y * case.t('dy')).sum(), [x, w])
    close(gx, case.t('gx'), 1e-5)
    close(gw, case.t('gw'), 1e-5)


@pytest.mark.parametrize('case', golden_io.load('modulated_conv2d'),
                         ids=lambda c: '-'.join(f'{k[0]}{v}' for k, v in c.meta.items()))
def test_modulated_conv2d(case):
    m = case.meta
    x, w, s = case.t('x', grad=True), case.t('w', grad=True), case.t('s', grad=True)
    y = stylegan2.modulated_conv2d(x, w, s, noise=case.get('noise'), up=m['up'], padding=m['kernel'] // 2,
                                   resample_filter=resample.make_filter([1, 3, 3, 1]), demodulate=m['demodulate'],
                                   flip_weight=(m['up'] == 1), fused_modconv=m['fused_modconv'])
    close(y, case.t('y'), 1e-5)
    gx, gw, gs = torch.autograd.grad((y * case.t('dy')).sum(), [x, w, s])
    close(gx, case.t('gx'), 1e-5)
    close(gw, case.t('gw'), 1e-5)
    close(gs, case.t('gs'), 1e-5)


@pytest.mark.parametrize('case', golden_io.load('misc_ops'), ids=lambda c: c.meta['op'])
def test_misc_ops(case):
    m = case.meta
    if m['op'] == 'mbstd':
        x = case.t('x', grad=True)
        y = pointwise.minibatch_std(x, m['group_size'], m['num_channels'])
        close(y, case.t('y'))
        gx, = torch.autograd.grad((y * case.t('dy')).sum(), x, create_graph=True)
        close(gx, case.t('gx'))
        ggx, = torch.autograd.grad((gx * case.t('v')).sum(), x)
        close(ggx, case.t('ggx'), 1e-5)
    else:
        a, b, c = case.t('a', grad=True), case.t('b', grad=True), case.t('c', grad=True)
        y = pointwise.fma(a, b, c)
        close(y, case.t('y'))
        ga, gb, gc = torch.autograd.grad((y * case.t('dy')).sum(), [a, b, c])
        close(ga, case.t('ga'))
        close(gb, case.t('gb'), 1e-5)
        close(gc, case.t('gc'), 1e-5)


def test_setup_filter_matches_fixture_filters():
    cases = golden_io.load('upfirdn2d')
    f44 = next(c for c in cases if c.meta['filter'] == 'f44').t('f')
    assert torch.equal(resample.make_filter([1, 3, 3, 1]), f44)
    f12 = next(c for c in cases if c.meta['filter'] == 'f12').t('f')
    assert f12.ndim == 1 and abs(float(f12.sum()) - 1) < 1e-6
