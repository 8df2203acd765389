"""The oracle's 16-bit storage emulation (oracle.stylegan2.low_precision) against the reference's float32 outputs: what
rounding every stored tensor of the four highest resolutions to bfloat16 does to a full-size generator image and to
a full-size discriminator's logits.  These distances are the noise floor that the GPU tests of the bfloat16 product
path are measured against; the emulation itself must stay within the bound the project states for 16-bit runs
(3e-2 relative L2 for images)."""
import contextlib
import zlib

import pytest
import torch

import golden_io
from oracle import stylegan2 as sg

FULL = golden_io.load('full_size_grads')


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


@pytest.mark.parametrize('case', [c for c in FULL if c.meta['tag'] == 'G256_affine_plus'], ids=lambda c: c.meta['tag'])
def test_rounded_generator_stays_within_the_16_bit_bound(case):
    m = case.meta
    p = {name: seeded_fill(name, shape) for name, shape in zip(m['names'], m['shapes'])}
    z = torch.randn(1, 512, generator=torch.Generator().manual_seed(m['z_seed']))
    cfg = sg.GeneratorConfig(img_resolution=m['res'], channel_base=m['channel_base'], conv_clamp=256, use_domain_modulation=True,
                             domain_modulation_parametrization=m['parametrization'])
    mid = m['res'] // 2
    errs = {}
    for name, ctx in (('float32', contextlib.nullcontext()), ('bfloat16', sg.low_precision(torch.bfloat16, m['res'] // 8))):
        with torch.no_grad(), ctx:
            img = sg.generator(p, z, torch.zeros(1, 0), cfg, noise_mode='const', fused_modconv=False)
        errs[name] = max(golden_io.rel_err(img[:, :, mid - 8:mid + 8, mid - 8:mid + 8], case.t('img_crop')),
                         golden_io.rel_err(img.double().sum(dim=[0, 1, 3]), case.t('img_row_sums')))
    assert errs['float32'] < 1e-4
    assert 1e-4 < errs['bfloat16'] < 3e-2          # rounding is visible, and within the stated bound


def test_rounding_is_straight_through_and_rounds_gradients():
    x = torch.tensor([1.0 + 2 ** -10, -3.0 - 2 ** -9], requires_grad=True)
    y = sg._rv(x, torch.bfloat16)
    assert y.tolist() == [1.0, -3.0] or y.tolist() == [1.0, -3.015625]      # to nearest, ties to even
    g, = torch.autograd.grad((y * torch.tensor([1.0 + 2 ** -10, 1.0])).sum(), x)
    assert g.tolist() == [1.0 + 2 ** -10, 1.0]                            # values rounded, gradient untouched
    x2 = torch.tensor([2.0], requires_grad=True)
    g2, = torch.autograd.grad((sg._rg(x2, torch.bfloat16) * (1.0 + 2 ** -10)).sum(), x2)
    assert g2.tolist() == [1.0]                                           # gradient rounded on the way back


def test_low_precision_leaves_the_low_resolutions_alone():
    case = [c for c in FULL if c.meta['tag'] == 'D256'][0]
    m = case.meta
    p = {name: seeded_fill(name, shape) for name, shape in zip(m['names'], m['shapes'])}
    x = torch.randn(m['batch'], 3, m['res'], m['res'], generator=torch.Generator().manual_seed(m['img_seed'])).clamp_(-1, 1)
    cfg = sg.DiscriminatorConfig(img_resolution=m['res'], channel_base=m['channel_base'], conv_clamp=256, mbstd_group_size=m['mbstd_group_size'])
    with torch.no_grad():
        plain = sg.discriminator(p, x, torch.zeros(m['batch'], 0), cfg)
        with sg.low_precision(torch.bfloat16, from_resolution=2 * m['res']):       # no block is that large
            same = sg.discriminator(p, x, torch.zeros(m['batch'], 0), cfg)
        with sg.low_precision(torch.bfloat16, from_resolution=m['res'] // 8):
            rounded = sg.discriminator(p, x, torch.zeros(m['batch'], 0), cfg)
    assert torch.equal(plain, same)
    scale = float(plain.abs().max())
    assert 0 < float((rounded - plain).abs().max()) < 0.1 * scale
    assert float((plain - case.t('logits')).abs().max()) < 1e-4 * max(scale, 1e-3) + 1e-5
