"""The functional CPU oracle for layers / generator / discriminator against the
reference's own outputs (tests/golden/{layers,networks,full_size}.npz)."""
import math
import zlib

import numpy as np
import pytest
import torch

import golden_io
from oracle import stylegan2 as sg

TOL = 2e-5


def close(a, b, tol=TOL):
    scale = max(float(b.detach().abs().max()), 1.0)
    assert golden_io.max_err(a, b) <= tol * scale, (golden_io.max_err(a, b), scale)


class LayerCfg:
    conv_clamp = 256
    use_noise = True


@pytest.mark.parametrize('case', golden_io.load('layers'), ids=lambda c: f"{c.meta['tag']}-{c.meta['kind']}")
def test_layers(case):
    m = case.meta
    p = {k: v.requires_grad_(v.is_floating_point()) for k, v in case.group('sd.').items()}
    kinds = sg.split_parametrization(m['use_domain_modulation'], m['parametrization'])
    x, w = case.t('x'), case.t('w')
    if m['kind'] == 'torgb':
        y = sg.torgb_layer(p, '', x, w, LayerCfg, kinds, fused_modconv=False)
        yf = sg.torgb_layer(p, '', x, w, LayerCfg, kinds, fused_modconv=True)
    else:
        up = 2 if m['kind'] == 'conv_up' else 1
        y = sg.synthesis_layer(p, '', x, w, LayerCfg, kinds, up=up, noise_mode='const', fused_modconv=False)
        yf = sg.synthesis_layer(p, '', x, w, LayerCfg, kinds, up=up, noise_mode='const', fused_modconv=True,
                                gain=math.sqrt(0.5))
    close(y, case.t('y'))
    close(yf, case.t('y_fused'))
    want = case.group('g.')
    names = [k for k in want]
    grads = torch.autograd.grad((y * case.t('dy')).sum(), [p[k] for k in names], allow_unused=True)
    for k, g in zip(names, grads):
        close(g if g is not None else torch.zeros_like(p[k]), want[k], 5e-5)


def gcfg(m):
    return sg.GeneratorConfig(img_resolution=m['res'], z_dim=m['z_dim'], w_dim=m['w_dim'], c_dim=m['c_dim'],
                              channel_base=m['channel_base'], channel_max=m['channel_max'], conv_clamp=m['conv_clamp'],
                              mapping_layers=m['mapping_layers'], use_domain_modulation=m['parametrization'] is not None,
                              domain_modulation_parametrization=m['parametrization'] or '')


NETS = golden_io.load('networks')


@pytest.mark.parametrize('case', [c for c in NETS if c.meta['kind'] == 'G'], ids=lambda c: c.meta['tag'])
def test_generator(case):
    m = case.meta
    cfg = gcfg(m)
    p = case.group('sd.')
    for k in m['trainable']:
        p[k].requires_grad_(True)
    z, c = case.t('z'), case.t('c')
    ws = sg.mapping(p, z, c, cfg)
    close(sg.w_avg_update(p['mapping.w_avg'], ws[:, 0], 0.995), case.t('w_avg_after'))
    img = sg.synthesis(p, ws, cfg, noise_mode='const', fused_modconv=False)
    close(img, case.t('img'))
    grads = torch.autograd.grad((img * case.t('dy')).sum(), [p[k] for k in m['trainable']])
    want = case.group('g.')
    for k, g in zip(m['trainable'], grads):
        close(g, want[k], 1e-4)
    with torch.no_grad():
        p['mapping.w_avg'] = case.t('w_avg_after')
        close(sg.generator(p, z, c, cfg, truncation_psi=0.7, truncation_cutoff=3, noise_mode='const',
                           fused_modconv=True), case.t('img_eval'))
        close(sg.generator(p, z, c, cfg, noise_mode='none', fused_modconv=True), case.t('img_none'))


@pytest.mark.parametrize('case', [c for c in NETS if c.meta['kind'] == 'D'], ids=lambda c: c.meta['tag'])
def test_discriminator(case):
    m = case.meta
    cfg = sg.DiscriminatorConfig(img_resolution=m['res'], c_dim=m['c_dim'], architecture=m['architecture'],
                                 channel_base=m['channel_base'], channel_max=m['channel_max'],
                                 conv_clamp=m['conv_clamp'], mbstd_group_size=m['mbstd_group_size'])
    p = {k: v.requires_grad_(v.is_floating_point() and not k.endswith('resample_filter'))
         for k, v in case.group('sd.').items()}
    img = case.t('img', grad=True)
    logits = sg.discriminator(p, img, case.t('c'), cfg)
    close(logits, case.t('logits'))
    want, want2 = case.group('g.'), case.group('g2.')
    names = list(want)
    grads = torch.autograd.grad(logits.sum(), [img] + [p[k] for k in names], create_graph=True)
    close(grads[0], case.t('gimg'))
    for k, g in zip(names, grads[1:]):
        close(g, want[k], 1e-4)
    r1 = grads[0].square().sum()
    close(r1, case.t('r1'), 1e-4)
    g2 = torch.autograd.grad(r1, [p[k] for k in names], allow_unused=True)
    for k, g in zip(names, g2):
        close(g if g is not None else torch.zeros_like(p[k]), want2[k], 2e-4)


def seeded_fill(name, shape):
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7fffffff)
    t = torch.randn(list(shape), generator=gen)
    if name.endswith('noise_strength'):
        t = t * 0.1
    elif name.endswith('.bias') and '.affine.' in name:
        t = t * 0.1 + 1
    elif name.endswith('.bias') or 'weights_offset' in name or name.endswith('w_avg'):
        t = t * 0.1
    return t


def test_full_size_generator_256():
    case, = golden_io.load('full_size')
    m = case.meta
    p = {n: seeded_fill(n, s) for n, s in zip(m['names'], m['shapes'])}
    cfg = sg.GeneratorConfig(img_resolution=256, channel_base=16384, conv_clamp=256, use_domain_modulation=True,
                             domain_modulation_parametrization=m['parametrization'])
    with torch.no_grad():
        img = sg.generator(p, case.t('z'), torch.zeros(1, 0), cfg, noise_mode='const', fused_modconv=False)
    close(img[:, :, 120:136, 120:136], case.t('crop'), 1e-4)
    rows = img.double().sum(dim=[0, 1, 3])
    assert golden_io.rel_err(rows, case.t('row_sums')) < 1e-4
    assert abs(float(img.double().abs().sum()) / float(case.np('abs_sum')) - 1) < 1e-5


def test_generator_512_affine_plus_against_reference_gradients():
    """The 512 x 512 Affine+ generator (the discriminator-heavy configuration's generator), batch 1: the oracle's
    image and the gradient of every trainable parameter against the reference's float32 run.  The style gradients
    are cancelling sums (ill-conditioned in float32): a tensor is held to 2e-3 or three times the reference's own
    distance from the float64 evaluation stored beside it, whichever is larger."""
    case = [c for c in golden_io.load('full_size_grads_more') if c.meta['tag'] == 'G512_affine_plus'][0]
    m = case.meta
    p = {n: seeded_fill(n, s) for n, s in zip(m['names'], m['shapes'])}
    for k in m['trainable']:
        p[k].requires_grad_(True)
    cfg = sg.GeneratorConfig(img_resolution=m['res'], channel_base=m['channel_base'], conv_clamp=256, use_domain_modulation=True,
                             domain_modulation_parametrization=m['parametrization'])
    z = torch.randn(1, 512, generator=torch.Generator().manual_seed(m['z_seed']))
    dy = torch.randn(1, 3, m['res'], m['res'], generator=torch.Generator().manual_seed(m['dy_seed']))
    img = sg.generator(p, z, torch.zeros(1, 0), cfg, noise_mode='const', fused_modconv=False)
    mid = m['res'] // 2
    assert golden_io.rel_err(img.detach()[:, :, mid - 8:mid + 8, mid - 8:mid + 8], case.t('img_crop')) < 1e-4
    assert golden_io.rel_err(img.detach().double().sum(dim=[0, 1, 3]), case.t('img_row_sums')) < 1e-4
    grads = torch.autograd.grad((img * dy).sum(), [p[k] for k in m['trainable']])
    for k, g in zip(m['trainable'], grads):
        whole = 'g.' + k in case
        got = g if whole else g.flatten()[::61]
        ref, exact = case.t(('g.' if whole else 'gs.') + k), case.t(('xg.' if whole else 'xgs.') + k)
        assert golden_io.rel_err(got, exact) < max(2e-3, 3 * golden_io.rel_err(ref, exact)), k
