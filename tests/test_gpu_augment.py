"""GPU parity of the augmentation pipeline against the reference's AugmentPipe (tests/golden/augment.npz, written
by tools/gen_golden.py with every random draw pinned to a percentile): outputs, image gradients and the
gradient of an R1-style penalty through the geometric resampling.  float32: 1e-4 of the tensor maximum for
outputs and first-order gradients, 1e-3 for the second-order one."""
import numpy as np
import pytest
import torch

import golden_io

pytestmark = pytest.mark.gpu
DEV = 'cuda'
CASES = golden_io.load('augment')


def close(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-3)
    assert float((a - b).abs().max()) <= tol * scale, (what, float((a - b).abs().max()), scale)


def test_buffers_match_reference():
    from ga_gan_amd.training import augment
    case = [c for c in CASES if c.meta['spec'] == 'buffers'][0]
    pipe = augment.AugmentPipe()
    np.testing.assert_allclose(pipe.Hz_geom.numpy(), case.np('Hz_geom'), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(pipe.Hz_fbank.numpy(), case.np('Hz_fbank'), rtol=1e-6, atol=1e-8)
    assert sorted(pipe.state_dict()) == ['Hz_fbank', 'Hz_geom', 'p']


@pytest.mark.parametrize('case', [c for c in CASES if c.meta['spec'] != 'buffers'],
                         ids=lambda c: f"{c.meta['spec']}-{'x'.join(map(str, c.meta['shape']))}-q{c.meta['q']}")
def test_pipe_matches_reference(case):
    from ga_gan_amd.training import augment
    from ga_gan_amd.torch_utils.ops import grid_sample_gradfix
    m = case.meta
    pipe = augment.AugmentPipe(**m['kwargs']).to(DEV)
    old, grid_sample_gradfix.enabled = grid_sample_gradfix.enabled, True
    try:
        x = case.t('x', DEV, grad=True)
        y = pipe(x, debug_percentile=m['q'])
        close(y, case.t('y'), 1e-4, 'y')
        gx, = torch.autograd.grad((y * case.t('dy', DEV)).sum(), x)
        close(gx, case.t('gx'), 1e-4, 'gx')
        if 'r1' in case:
            gy, = torch.autograd.grad(pipe(x, debug_percentile=m['q']).square().sum(), x, create_graph=True)
            r1, = torch.autograd.grad(gy.square().sum(), x)
            close(r1, case.t('r1'), 1e-3, 'r1')
    finally:
        grid_sample_gradfix.enabled = old


def test_strength_zero_is_identity_and_random_mode_runs():
    from ga_gan_amd.training import augment
    torch.manual_seed(0)
    kwargs = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1,
                  hue=1, saturation=1, imgfilter=1, noise=1, cutout=1)
    pipe = augment.AugmentPipe(**kwargs).to(DEV)
    x = torch.randn(4, 3, 32, 32, device=DEV)
    pipe.p.copy_(torch.zeros([]))
    close(pipe(x), x, 2e-4, 'p = 0')       # (the resampling chain still runs, on identity transforms)
    pipe.p.copy_(torch.ones([]) * 0.6)
    y = pipe(x)
    assert y.shape == x.shape and bool(torch.isfinite(y).all())
    assert float((y - x).abs().max()) > 1e-2


def test_trainer_with_ada_adjusts_strength_and_regularises_through_the_pipe():
    """A few iterations of the trainer with the bgc pipe in front of D: all four phases run (R1 differentiates twice
    through the resampling), and the strength p moves by batch * interval / (ada_kimg * 1000) per adjustment."""
    from ga_gan_amd.training import training_loop
    opts = training_loop.setup_training_options(cfg='paper256', resolution=32, batch_size=4, **training_loop.AFFINE_PLUS)
    for k in ('G_kwargs', 'D_kwargs'):
        kw = opts[k].get('synthesis_kwargs', opts[k])
        kw['channel_base'], kw['channel_max'] = 512, 32
    pipe = training_loop.make_augment_pipe('bgc', 0.25, DEV)
    trainer = training_loop.Trainer(random_seed=0, augment_pipe=pipe, ada_target=0.6, ada_interval=2, ada_kimg=1,
                                    G_reg_interval=2, D_reg_interval=2, **opts)
    gen = torch.Generator(device=DEV).manual_seed(1)
    seen = [float(pipe.p)]
    for _ in range(4):
        img, c = trainer.synthetic_batch(gen)
        trainer.step(img, c)
        seen.append(float(pipe.p))
    step = 4 * 2 / 1000
    assert seen[1] == seen[0] and seen[3] == seen[2]                      # adjusted every second iteration only
    assert abs(abs(seen[2] - seen[0]) - step) < 1e-6 and abs(abs(seen[4] - seen[2]) - step) < 1e-6
    assert all(bool(torch.isfinite(p).all()) for p in trainer.D.parameters())
    assert sorted(training_loop.AUGPIPE_SPECS) == ['bg', 'bgc', 'bgcf', 'bgcfn', 'bgcfnc', 'blit', 'color', 'cutout', 'filter',
                                                   'geom', 'noise']
