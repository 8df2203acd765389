"""GPU parity of the product networks / loss (HIP op path) against the reference's
outputs stored in tests/golden and against the CPU oracle.  Run with -m gpu.

float32 tolerance: 1e-4 of the tensor's max magnitude for activations, 1e-3 for
parameter gradients of whole networks (sums over up to 1e5 products in a different
order than the CPU path); images additionally within 1e-3 relative L2, the bound
the project states for generated images.
"""
import zlib

import pytest
import torch

import golden_io

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def close(a, b, tol, what=''):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-3)
    err = float((a - b).abs().max())
    assert err <= tol * scale, (what, err, scale)


def networks():
    from ga_gan_amd.training import networks as n
    return n


@pytest.mark.parametrize('case', golden_io.load('layers'), ids=lambda c: f"{c.meta['tag']}-{c.meta['kind']}")
def test_layers(case):
    n = networks()
    m = case.meta
    kw = dict(w_dim=16, resolution=16, conv_clamp=256, use_domain_modulation=m['use_domain_modulation'],
              domain_modulation_parametrization=m['parametrization'], generator_requires_grad_parts=tuple(m['parts']))
    if m['kind'] == 'torgb':
        layer = n.ToRGBLayer(8, 3, **kw)
    else:
        layer = n.SynthesisLayer(8, 6, up=2 if m['kind'] == 'conv_up' else 1, **kw)
    layer.load_state_dict(case.group('sd.'), strict=True)
    layer.to(DEV)
    x, w = case.t('x', DEV), case.t('w', DEV)
    extra = {} if m['kind'] == 'torgb' else dict(gain=0.5 ** 0.5)
    y = layer(x, w, fused_modconv=False, noise_mode='const')
    yf = layer(x, w, fused_modconv=True, noise_mode='const', **extra)
    close(y, case.t('y'), 1e-4, 'y')
    close(yf, case.t('y_fused'), 1e-4, 'y_fused')
    want = case.group('g.')
    params = dict(layer.named_parameters())
    assert sorted(params) == sorted(want)
    grads = torch.autograd.grad((y * case.t('dy', DEV)).sum(), [params[k] for k in want], allow_unused=True)
    for k, g in zip(want, grads):
        close(g if g is not None else torch.zeros_like(params[k]), want[k], 2e-4, k)


NETS = golden_io.load('networks')


def build_G(m):
    n = networks()
    sk = dict(channel_base=m['channel_base'], channel_max=m['channel_max'], num_fp16_res=0, conv_clamp=m['conv_clamp'],
              generator_requires_grad_parts=list(m['parts']))
    if m['parametrization'] is not None:
        sk.update(use_domain_modulation=True, domain_modulation_parametrization=m['parametrization'])
    return n.Generator(z_dim=m['z_dim'], c_dim=m['c_dim'], w_dim=m['w_dim'], img_resolution=m['res'], img_channels=3,
                       mapping_kwargs=dict(num_layers=m['mapping_layers']), synthesis_kwargs=sk)


def build_D(m, **extra):
    n = networks()
    return n.Discriminator(c_dim=m['c_dim'], img_resolution=m['res'], img_channels=3, architecture=m.get('architecture', 'resnet'),
                           channel_base=m['channel_base'], channel_max=m['channel_max'], num_fp16_res=0,
                           conv_clamp=m['conv_clamp'], epilogue_kwargs=dict(mbstd_group_size=m['mbstd_group_size']), **extra)


@pytest.mark.parametrize('case', [c for c in NETS if c.meta['kind'] == 'G'], ids=lambda c: c.meta['tag'])
def test_generator(case):
    from ga_gan_amd.training import training_loop
    m = case.meta
    G = build_G(m)
    G.load_state_dict(case.group('sd.'), strict=True)
    G.to(DEV).train()
    training_loop.set_requires_grad(G, list(m['parts']))
    assert sorted(k for k, p in G.named_parameters() if p.requires_grad) == sorted(m['trainable'])
    z, c = case.t('z', DEV), case.t('c', DEV)
    img = G(z, c, noise_mode='const')
    close(img, case.t('img'), 1e-4, 'img')
    assert golden_io.rel_err(img, case.t('img')) < 1e-3
    close(G.mapping.w_avg, case.t('w_avg_after'), 1e-5, 'w_avg')
    params = dict(G.named_parameters())
    grads = torch.autograd.grad((img * case.t('dy', DEV)).sum(), [params[k] for k in m['trainable']])
    want = case.group('g.')
    for k, g in zip(m['trainable'], grads):
        close(g, want[k], 1e-3, k)
    G.eval()
    with torch.no_grad():
        close(G(z, c, noise_mode='const', truncation_psi=0.7, truncation_cutoff=3), case.t('img_eval'), 1e-4, 'eval')
        close(G(z, c, noise_mode='none'), case.t('img_none'), 1e-4, 'none')


@pytest.mark.parametrize('case', [c for c in NETS if c.meta['kind'] == 'D'], ids=lambda c: c.meta['tag'])
def test_discriminator(case):
    m = case.meta
    extra = dict(block_kwargs=dict(freeze_layers=3)) if m['tag'] == 'D32_freeze' else {}
    D = build_D(m, **extra)
    D.load_state_dict(case.group('sd.'), strict=True)
    D.to(DEV).train().requires_grad_(True)
    img = case.t('img', DEV, grad=True)
    logits = D(img, case.t('c', DEV))
    close(logits, case.t('logits'), 1e-4, 'logits')
    want, want2 = case.group('g.'), case.group('g2.')
    params = dict(D.named_parameters())
    assert sorted(params) == sorted(want)
    names = list(want)
    grads = torch.autograd.grad(logits.sum(), [img] + [params[k] for k in names], create_graph=True)
    close(grads[0], case.t('gimg'), 1e-4, 'gimg')
    for k, g in zip(names, grads[1:]):
        close(g, want[k], 1e-3, k)
    r1 = grads[0].square().sum()
    close(r1, case.t('r1'), 1e-3, 'r1')
    g2 = torch.autograd.grad(r1, [params[k] for k in names], allow_unused=True)
    for k, g in zip(names, g2):
        close(g if g is not None else torch.zeros_like(params[k]), want2[k], 2e-3, 'g2.' + k)


@pytest.mark.parametrize('case', golden_io.load('loss'), ids=lambda c: c.meta['tag'])
def test_loss_phases(case):
    from ga_gan_amd.training import loss as loss_module
    from ga_gan_amd.training import training_loop
    m = case.meta
    m = dict(m, c_dim=0)
    G, D = build_G(m), build_D(m)
    G.load_state_dict(case.group('G.'), strict=True)
    D.load_state_dict(case.group('D.'), strict=True)
    G.to(DEV).train().requires_grad_(False)
    D.to(DEV).train().requires_grad_(False)
    loss = loss_module.StyleGAN2Loss(device=torch.device(DEV), G_mapping=G.mapping, G_synthesis=G.synthesis, D=D,
                                     style_mixing_prob=0, r1_gamma=m['r1_gamma'], pl_batch_shrink=m['pl_batch_shrink'],
                                     pl_decay=m['pl_decay'], pl_weight=m['pl_weight'])
    loss.synthesis_kwargs = dict(noise_mode='const')
    z, c, real = case.t('z', DEV), case.t('c', DEV), case.t('real', DEV)
    for phase in ('Gmain', 'Greg', 'Dmain', 'Dreg'):
        module, parts = (G, list(m['parts'])) if phase[0] == 'G' else (D, ['all'])
        G.zero_grad(set_to_none=True)
        D.zero_grad(set_to_none=True)
        G.requires_grad_(False)
        D.requires_grad_(False)
        training_loop.set_requires_grad(module, parts)
        close(G.mapping.w_avg, case.t(f'{phase}.w_avg_before'), 1e-5, 'w_avg before ' + phase)
        loss.accumulate_gradients(phase=phase, real_img=real, real_c=c, gen_z=z, gen_c=c, sync=True,
                                  gain=m['gains'][phase], pl_noise_unit=case.t('pl_noise_unit', DEV))
        want = case.group(f'{phase}.g.')
        got = {k: p.grad for k, p in module.named_parameters() if p.grad is not None}
        assert sorted(got) == sorted(want), phase
        for k in want:
            close(got[k], want[k], 2e-3, f'{phase} {k}')
        close(G.mapping.w_avg, case.t(f'{phase}.w_avg_after'), 1e-5, 'w_avg after ' + phase)
        close(loss.pl_mean, case.t(f'{phase}.pl_mean_after'), 1e-4, 'pl_mean ' + phase)
        module.requires_grad_(False)


@pytest.mark.parametrize('batch,group', [(8, 4), (8, 8), (12, 4), (4, 8), (6, None)])
def test_dmain_single_pass_equals_two_passes(batch, group):
    """Dmain with both halves in one discriminator pass (rows interleaved so that the minibatch-std groups are the
    ones of two separate passes) against the reference's two passes: logits statistics and every gradient."""
    from ga_gan_amd.training import loss as loss_module
    from ga_gan_amd.torch_utils import training_stats
    case = golden_io.load('loss')[0]
    m = dict(case.meta, c_dim=0)
    G, D = build_G(m), build_D(dict(m, mbstd_group_size=group))
    G.load_state_dict(case.group('G.'), strict=True)
    D.load_state_dict(case.group('D.'), strict=True)
    G.to(DEV).train().requires_grad_(False)
    D.to(DEV).train().requires_grad_(True)
    loss = loss_module.StyleGAN2Loss(device=torch.device(DEV), G_mapping=G.mapping, G_synthesis=G.synthesis, D=D, style_mixing_prob=0)
    loss.synthesis_kwargs = dict(noise_mode='const')
    g = torch.Generator().manual_seed(batch)
    z = torch.randn([batch, G.z_dim], generator=g).to(DEV)
    real = torch.randn([batch, 3, m['res'], m['res']], generator=g).clamp(-1, 1).to(DEV)
    c = torch.zeros([batch, 0], device=DEV)
    results = {}
    for merged in (False, True):
        loss.merge_D_passes = merged
        D.zero_grad(set_to_none=True)
        w_avg = G.mapping.w_avg.clone()
        loss.accumulate_gradients(phase='Dmain', real_img=real, real_c=c, gen_z=z, gen_c=c, sync=True, gain=1)
        G.mapping.w_avg.copy_(w_avg)
        results[merged] = {k: p.grad.clone() for k, p in D.named_parameters()}
    expect_merge = group is not None and group <= batch and batch % group == 0
    assert (loss._merged_groups(batch, batch) is not None) == expect_merge
    for k in results[False]:
        close(results[True][k], results[False][k], 1e-4, k)


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


def full_size_generator(num_fp16_res):
    from ga_gan_amd.training import training_loop
    opts = training_loop.setup_training_options(cfg='paper256', resolution=256, batch_size=4, **training_loop.AFFINE_PLUS)
    kw = opts['G_kwargs']
    kw['synthesis_kwargs'].update(num_fp16_res=num_fp16_res, conv_clamp=256)
    G = networks().Generator(**kw)
    case, = golden_io.load('full_size')
    sd = G.state_dict()
    for name, shape in zip(case.meta['names'], case.meta['shapes']):
        assert list(sd[name].shape) == shape, name
        sd[name] = seeded_fill(name, shape)
    G.load_state_dict(sd, strict=True)
    return G.to(DEV).train(), case


def test_full_size_generator_256_fp32():
    """FFHQ-256 Affine+ generator, batch 1, float32: within 1e-3 relative L2 of the reference's image."""
    G, case = full_size_generator(num_fp16_res=0)
    with torch.no_grad():
        img = G(case.t('z', DEV), torch.zeros(1, 0, device=DEV), noise_mode='const')
    crop = img[:, :, 120:136, 120:136]
    assert golden_io.rel_err(crop, case.t('crop')) < 1e-3
    assert golden_io.rel_err(img.double().sum(dim=[0, 1, 3]), case.t('row_sums')) < 1e-3
    assert abs(float(img.double().abs().sum()) / float(case.np('abs_sum')) - 1) < 1e-4


def test_full_size_generator_256_bf16():
    """Same generator with the four highest resolutions in bfloat16 (the bench configuration): the image
    stays within 3e-2 relative L2 of the float32 reference image (bf16 keeps 8 significant bits per layer)."""
    G, case = full_size_generator(num_fp16_res=4)
    with torch.no_grad():
        img = G(case.t('z', DEV), torch.zeros(1, 0, device=DEV), noise_mode='const')
    assert img.dtype == torch.float32
    assert golden_io.rel_err(img[:, :, 120:136, 120:136], case.t('crop')) < 3e-2
    assert golden_io.rel_err(img.double().sum(dim=[0, 1, 3]), case.t('row_sums')) < 3e-2


FULL = golden_io.load('full_size_grads') + golden_io.load('full_size_grads_more')     # (the second: 1024 full fine-tune, 512 Affine+)


def _dm(m):
    """Domain-modulation arguments of a full-size generator case (the first bundle's cases all use it)."""
    on = m.get('use_domain_modulation', True)
    return dict(use_domain_modulation=on, domain_modulation_parametrization=m['parametrization'] if on else 'multiplicative')


def fill_seeded(module, m):
    sd = module.state_dict()
    for name, shape in zip(m['names'], m['shapes']):
        assert list(sd[name].shape) == shape, name
        sd[name] = seeded_fill(name, shape)
    module.load_state_dict(sd, strict=True)


def check_summary(case, prefix, t, tol, crop):
    assert golden_io.rel_err(t[:, :, crop[0]:crop[1], crop[0]:crop[1]], case.t(prefix + 'crop')) < tol, prefix + 'crop'
    assert golden_io.rel_err(t.double().sum(dim=[0, 1, 3]), case.t(prefix + 'row_sums')) < tol, prefix + 'row_sums'
    assert abs(float(t.double().abs().sum()) / float(case.np(prefix + 'abs_sum')) - 1) < tol, prefix + 'abs_sum'


@pytest.mark.parametrize('num_fp16_res', [0, 4], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('case', [c for c in FULL if c.meta['kind'] == 'G'], ids=lambda c: c.meta['tag'])
def test_full_size_generator_gradients(case, num_fp16_res):
    """Full-size generators (256 Affine+, 1024 AffineLight+, 1024 with every synthesis parameter trainable and no
    domain modulation, 512 Affine+), batch 1, training mode, on seeded parameters: image and the gradient of every
    trainable parameter.

    The image is held to 1e-3 relative L2 of the reference's (float32) and 3e-2 (bfloat16 in the four highest
    resolutions).  The style gradients are sums of 1e5 .. 1e6 signed terms per channel that cancel to a small
    remainder, which makes them ill-conditioned in float32: the reference's own float32 result lies 2e-4 .. 5e-3
    (relative L2, per tensor) from the same gradient evaluated in float64 (fixture entries 'xg.'), and two CPU float32
    evaluations of the same graph differ from each other by as much.  Each tensor is therefore compared with the
    float64 gradient, with a tolerance of a floor or three times the reference's own distance from it, whichever is
    larger.  The floor is 2e-3 up to 256 pixels and 4e-3 above: the product's float32 layers multiply on the matrix
    cores with 22 of float32's 24 significand bits (each operand as two half-precision pieces), four times the
    rounding of true float32 per product, and the 1024 generator passes a gradient through eighteen such layers
    (measured: 2.0e-3 .. 2.7e-3 on its 4 .. 16 pixel layers, where the reference's float32 run is at 7e-4).
    One-element tensors (noise_strength: a single sum over all pixels and channels that cancels to a small
    remainder, which multiplies whatever error its terms arrive with) get twice the floor or twelve times the
    reference's distance (seen: up to 9 times, and 1.03 floors after a change of summation order in one kernel).
    bfloat16 runs (activations and gradients pass through up to eight bf16 layers, twice) get 0.2 per tensor, except
    one-element tensors and the tensors where the reference itself is off by 2e-3 or more; all gradients taken as
    one vector must lie within the floor (float32) and 0.1 (bfloat16)."""
    from ga_gan_amd.training import training_loop
    m = case.meta
    G = networks().Generator(
        z_dim=512, c_dim=0, w_dim=512, img_resolution=m['res'], img_channels=3, mapping_kwargs=dict(num_layers=8),
        synthesis_kwargs=dict(channel_base=m['channel_base'], channel_max=512, num_fp16_res=num_fp16_res, conv_clamp=256,
                              generator_requires_grad_parts=m['parts'], **_dm(m)))
    fill_seeded(G, m)
    G.to(DEV).train()
    training_loop.set_requires_grad(G, list(m['parts']))
    assert sorted(k for k, p in G.named_parameters() if p.requires_grad) == sorted(m['trainable'])
    z = torch.randn(1, 512, generator=torch.Generator().manual_seed(m['z_seed'])).to(DEV)
    img = G(z, torch.zeros(1, 0, device=DEV), noise_mode='const')
    assert img.dtype == torch.float32
    mid = m['res'] // 2
    check_summary(case, 'img_', img.detach(), 1e-3 if num_fp16_res == 0 else 3e-2, (mid - 8, mid + 8))
    dy = torch.randn(img.shape, generator=torch.Generator().manual_seed(m['dy_seed'])).to(DEV)
    params = dict(G.named_parameters())
    grads = torch.autograd.grad((img * dy).sum(), [params[k] for k in m['trainable']])
    floor = 2e-3 if m['res'] <= 256 else 4e-3
    worst, num, den = {}, 0.0, 0.0
    for k, g in zip(m['trainable'], grads):
        whole = 'g.' + k in case
        got = g if whole else g.flatten()[::61]
        ref, exact = case.t(('g.' if whole else 'gs.') + k), case.t(('xg.' if whole else 'xgs.') + k)
        ref_err = golden_io.rel_err(ref, exact)
        err = golden_io.rel_err(got, exact)
        num += float((got.detach().double().cpu() - exact.double()).square().sum())
        den += float(exact.double().square().sum())
        if num_fp16_res == 0:
            tol = max(floor, 3 * ref_err) if g.numel() > 1 else max(2 * floor, 12 * ref_err)
        elif g.numel() == 1 or ref_err >= 2e-3:
            tol = float('inf')                                 # (the worst-conditioned tensors count in the total only)
        else:
            tol = 0.2
        if err >= tol:
            worst[k] = (err, tol)
    assert not worst, worst
    assert (num / den) ** 0.5 < (floor if num_fp16_res == 0 else 0.1)      # all trainable gradients as one vector


@pytest.mark.parametrize('num_fp16_res', [0, 4], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('case', [c for c in FULL if c.meta['kind'] == 'D'], ids=lambda c: c.meta['tag'])
def test_full_size_discriminator(case, num_fp16_res):
    """Full-size discriminators at 256 / 512 / 1024 on seeded parameters and images: logits, the image gradient of
    their sum, every bias gradient and the norm of every weight gradient against the reference's float32 run.
    float32: 1e-3 (relative L2; logits 1e-4 of their magnitude).  bfloat16 in the four highest resolutions: 0.1 for
    the logits (a sum of O(1) terms that cancels to 0.06 .. 0.5 on these random weights) and 0.25 for gradients
    (eight bf16 layers forward and back)."""
    m = case.meta
    D = networks().Discriminator(c_dim=0, img_resolution=m['res'], img_channels=3, channel_base=m['channel_base'],
                                 channel_max=512, num_fp16_res=num_fp16_res, conv_clamp=256,
                                 epilogue_kwargs=dict(mbstd_group_size=m['mbstd_group_size']))
    fill_seeded(D, m)
    D.to(DEV).train().requires_grad_(True)
    n, res = m['batch'], m['res']
    img = torch.randn(n, 3, res, res, generator=torch.Generator().manual_seed(m['img_seed'])).clamp_(-1, 1).to(DEV).requires_grad_(True)
    logits = D(img, torch.zeros(n, 0, device=DEV))
    tol_logits, tol = (1e-4, 1e-3) if num_fp16_res == 0 else (0.1, 0.25)
    close(logits, case.t('logits'), tol_logits, 'logits')
    names = [k for k, _ in D.named_parameters()]
    grads = torch.autograd.grad(logits.sum(), [img] + list(D.parameters()))
    mid = res // 2
    check_summary(case, 'gimg_', grads[0].detach(), tol, (mid - 8, mid + 8))
    for k, g in zip(names, grads[1:]):
        if k.endswith('.bias'):
            assert golden_io.rel_err(g, case.t('g.' + k)) < tol, k
        else:
            assert abs(float(g.double().norm()) / float(case.np('gnorm.' + k)) - 1) < tol, k


STEP_SHAPES = [  # per-GPU shapes of the larger bench configurations
    ('c3', 'stylegan2', 1024, 16, 'AFFINE_LIGHT_PLUS'),
    ('c4', 'stylegan2', 1024, 8, 'FULL'),
    ('c5', 'stylegan2', 512, 8, 'AFFINE_PLUS'),
]


@pytest.mark.parametrize('tag,cfg,res,batch_gpu,par', STEP_SHAPES, ids=[s[0] for s in STEP_SHAPES])
def test_trainer_step_at_bench_shapes(tag, cfg, res, batch_gpu, par):
    """Three Trainer steps (Gmain, Greg, Dmain and Dreg all fire in the first one) at the per-GPU shapes of the
    1024 AffineLight+, 1024 full fine-tune and 512 Affine+ configurations: finite statistics, trainable generator
    parameters and the discriminator move, frozen generator parameters do not."""
    from ga_gan_amd.training import training_loop
    opts = training_loop.setup_training_options(cfg=cfg, resolution=res, batch_size=batch_gpu, num_gpus=1, batch_gpu=batch_gpu,
                                                glrate=0.02 if par != 'FULL' else None, **getattr(training_loop, par))
    trainer = training_loop.Trainer(rank=0, random_seed=0, **opts)
    before_G = {k: p.detach().clone() for k, p in trainer.G.named_parameters()}
    before_D = {k: p.detach().clone() for k, p in trainer.D.named_parameters()}
    gen = torch.Generator(device=trainer.device).manual_seed(1)
    real, real_c = trainer.synthetic_batch(gen)
    for _ in range(3):
        trainer.step(real, real_c)
    torch.cuda.synchronize()
    moved = [k for k, p in trainer.G.named_parameters() if not torch.equal(p.detach(), before_G[k])]
    assert moved, 'no generator parameter changed'
    if par != 'FULL':
        parts = getattr(training_loop, par)['generator_requires_grad_parts']
        training_loop.set_requires_grad(trainer.G, list(parts))
        allowed = {k for k, p in trainer.G.named_parameters() if p.requires_grad}
        assert set(moved) <= allowed, sorted(set(moved) - allowed)[:5]
    assert any(not torch.equal(p.detach(), before_D[k]) for k, p in trainer.D.named_parameters())
    for k, p in list(trainer.G.named_parameters()) + list(trainer.D.named_parameters()):
        assert bool(torch.isfinite(p).all()), k
    del trainer
    torch.cuda.empty_cache()


def test_dmain_at_1024_with_32_samples_takes_two_passes():
    """32 generated + 32 real images at 1024 x 1024 in one discriminator pass would make a first-layer tensor of 2**31
    elements, beyond the kernels' 32-bit indexing: the loss falls back to the reference's two passes and the step
    goes through (the one-pass form is used up to 16 + 16)."""
    from ga_gan_amd.training import training_loop
    opts = training_loop.setup_training_options(cfg='stylegan2', resolution=1024, batch_size=32, num_gpus=1, batch_gpu=32,
                                                glrate=0.02, **training_loop.AFFINE_LIGHT_PLUS)
    trainer = training_loop.Trainer(rank=0, random_seed=0, **opts)
    assert trainer.loss._merged_groups(32, 32) is None and trainer.loss._merged_groups(16, 16) == 4
    before = {k: p.detach().clone() for k, p in trainer.D.named_parameters()}
    real, real_c = trainer.synthetic_batch(torch.Generator(device=trainer.device).manual_seed(1))
    assert real.shape[0] == 32
    trainer.step(real, real_c)
    torch.cuda.synchronize()
    assert any(not torch.equal(p.detach(), before[k]) for k, p in trainer.D.named_parameters())
    for k, p in trainer.D.named_parameters():
        assert bool(torch.isfinite(p).all()), k
    del trainer
    torch.cuda.empty_cache()


@pytest.mark.parametrize('fuse', [True, False], ids=['fused', 'composed'])
@pytest.mark.parametrize('kernel_size', [1, 3])
def test_trainable_upsampling_conv2d_layer_gets_its_weight_gradient(kernel_size, fuse):
    """A trainable Conv2dLayer(up=2) - the skip convolution of a generator with architecture='resnet' - takes the
    paths that flip or transpose the prepared weight; those derived tensors must stay in the autograd graph.  Checked
    against the same layer written with torch's transposed convolution and a depthwise blur."""
    import torch.nn.functional as F
    n = networks()
    torch.manual_seed(kernel_size)
    layer = n.Conv2dLayer(16, 8, kernel_size=kernel_size, bias=False, up=2).to(DEV)
    x = torch.randn(2, 16, 16, 16, device=DEV, requires_grad=True)
    dy = torch.randn(2, 8, 32, 32, device=DEV)
    old = n.FUSE_LAYERS
    n.FUSE_LAYERS = fuse
    try:
        for _ in range(2):                       # the second pass meets whatever the first one cached
            layer.weight.grad = None
            x.grad = None
            y = layer(x)
            (y * dy).sum().backward()
            assert layer.weight.grad is not None and float(layer.weight.grad.abs().max()) > 0
            got = (y.detach().clone(), x.grad.clone(), layer.weight.grad.clone())
    finally:
        n.FUSE_LAYERS = old
    w2 = layer.weight.detach().double().requires_grad_()
    x2 = x.detach().double().requires_grad_()
    f1 = torch.tensor([1., 3., 3., 1.], dtype=torch.float64, device=DEV)
    f2 = torch.outer(f1, f1)
    f2 = (f2 / f2.sum() * 4)[None, None].repeat(8, 1, 1, 1)
    k = kernel_size
    if k == 1:   # 1x1: convolve, insert zeros, blur (reference conv2d_resample.py:110-114)
        c = F.conv2d(x2, w2 * layer.weight_gain)
        z = torch.zeros(2, 8, 32, 32, dtype=torch.float64, device=DEV)
        z[:, :, ::2, ::2] = c
        want = F.conv2d(F.pad(z, [2, 1, 2, 1]), f2, groups=8)
    else:        # transposed convolution with the unflipped weight, then the blur (reference conv2d_resample.py:126-137)
        c = F.conv_transpose2d(x2, (w2 * layer.weight_gain).transpose(0, 1), stride=2)
        want = F.conv2d(F.pad(c, [1, 1, 1, 1]), f2, groups=8)
    (want * dy.double()).sum().backward()
    assert golden_io.rel_err(got[0], want) < 1e-5
    assert golden_io.rel_err(got[1], x2.grad) < 1e-5
    assert golden_io.rel_err(got[2], w2.grad) < 1e-5


def test_conditional_trainer_takes_generated_labels_from_the_caller():
    """c_dim > 0: `Trainer.step` uses the caller's gen_c (one run per phase) and, without one, resamples the real
    batch's labels with replacement per phase (the reference draws every generated label from a random dataset item,
    training_loop.py:364-366)."""
    from ga_gan_amd.training import training_loop
    opts = training_loop.setup_training_options(cfg='paper256', resolution=32, batch_size=4, c_dim=5, **training_loop.FULL)
    for k in ('G_kwargs', 'D_kwargs'):
        kw = opts[k].get('synthesis_kwargs', opts[k])
        kw['channel_base'], kw['channel_max'] = 512, 32
    trainer = training_loop.Trainer(rank=0, random_seed=0, **opts)
    seen = []
    original = trainer.loss.accumulate_gradients

    def spy(**kw):
        seen.append((kw['phase'], kw['gen_c'].clone(), kw['real_c'].clone()))
        return original(**kw)

    trainer.loss.accumulate_gradients = spy
    real = torch.randint(0, 256, [4, 3, 32, 32], dtype=torch.uint8, device=DEV)
    real_c = torch.eye(5, device=DEV)[[0, 1, 2, 3]]
    gen_c = torch.eye(5, device=DEV)[torch.arange(16) % 5]
    trainer.step(real, real_c, gen_c=gen_c)
    assert [s[0] for s in seen] == ['Gmain', 'Greg', 'Dmain', 'Dreg']
    for i, (_, g_c, r_c) in enumerate(seen):
        assert torch.equal(g_c, gen_c[4 * i:4 * i + 4]) and torch.equal(r_c, real_c)
    seen.clear()
    torch.manual_seed(0)
    trainer.batch_idx = 16
    trainer.step(real, real_c)
    drawn = torch.cat([s[1] for s in seen])
    assert drawn.shape == (16, 5) and bool((drawn.sum(dim=1) == 1).all()) and float(drawn[:, 4].sum()) == 0
    assert not all(torch.equal(s[1], real_c) for s in seen)      # (resampled, not the real batch's order)


def test_derived_tensor_cache_follows_parameter_writes():
    """Frozen networks keep packed / cast weights between calls; an in-place write to any source parameter
    (optimizer step, load_state_dict) must invalidate them.  Checked against the same calls with the cache off,
    for float32 and bfloat16 blocks, before and after the writes, with and without requires_grad."""
    from ga_gan_amd.torch_utils import memo
    n = networks()
    torch.manual_seed(3)
    sk = dict(channel_base=2048, channel_max=64, num_fp16_res=2, conv_clamp=256, use_domain_modulation=True,
              domain_modulation_parametrization='out_in_additive,affine_out_in_5_1_additive',
              generator_requires_grad_parts=['all'])
    G = n.Generator(z_dim=32, c_dim=0, w_dim=32, img_resolution=64, img_channels=3, mapping_kwargs=dict(num_layers=2),
                    synthesis_kwargs=sk).to(DEV).train().requires_grad_(False)
    D = n.Discriminator(c_dim=0, img_resolution=64, img_channels=3, channel_base=2048, channel_max=64, num_fp16_res=2,
                        conv_clamp=256, epilogue_kwargs=dict(mbstd_group_size=2)).to(DEV).train().requires_grad_(False)
    z = torch.randn(4, 32, device=DEV)
    c = torch.zeros(4, 0, device=DEV)

    def run():
        img = G(z, c, noise_mode='const')
        return img, D(img, c)

    def check(what):
        memo.enabled = True
        got = run()
        got_again = run()          # second call: served from the cache
        memo.enabled = False
        try:
            want = run()
        finally:
            memo.enabled = True
        for a, b, w in zip(got, got_again, want):   # (split-K float atomics: equal up to summation order)
            close(a, w, 1e-4, what)
            close(b, w, 1e-4, what)

    check('initial')
    with torch.no_grad():
        for p in list(G.parameters()) + list(D.parameters()):
            p.add_(torch.randn_like(p) * 0.05)
    check('after in-place update')
    sd = {k: v + 0.01 for k, v in G.state_dict().items() if v.is_floating_point()}
    G.load_state_dict(sd, strict=False)
    check('after load_state_dict')
    # gradients still flow to parameters that ask for them (no constant is substituted for their weights)
    G.requires_grad_(True)
    img, logits = run()
    grads = torch.autograd.grad(logits.sum(), [p for p in G.parameters()], allow_unused=True)
    assert sum(g is not None and bool(g.abs().sum() > 0) for g in grads) > 20
    G.requires_grad_(False)
    check('after a pass with gradients')
    # trainable layers serve their prepared weights from the cache as well (values only; the gradient is attached
    # by the autograd function): gradients with the cache on, twice, against the cache off
    D.requires_grad_(True)
    d_params = list(D.parameters())

    def d_grads():
        img = G(z, c, noise_mode='const').detach()
        return torch.autograd.grad(D(img, c).sum(), d_params)

    got, got_again = d_grads(), d_grads()
    memo.enabled = False
    try:
        want = d_grads()
    finally:
        memo.enabled = True
    for a, b, w in zip(got, got_again, want):
        close(a, w, 1e-2, 'D gradient')   # (bfloat16 weight gradients: one unit in the last place is 4e-3)
        close(b, w, 1e-2, 'D gradient, cached')
    with torch.no_grad():
        for p in d_params:
            p.mul_(1.5)
    got = d_grads()
    memo.enabled = False
    try:
        want = d_grads()
    finally:
        memo.enabled = True
    for a, w in zip(got, want):
        close(a, w, 1e-2, 'D gradient after an update')   # (a stale weight would be off by tens of percent)


def test_trainer_snapshot_round_trip(tmp_path):
    """A snapshot written after two iterations and loaded into a fresh trainer restores networks, optimizer moments,
    counters and the path-length average; and the restored trainer keeps training."""
    from ga_gan_amd.training import training_loop
    opts = training_loop.setup_training_options(cfg='paper256', resolution=32, batch_size=4, **training_loop.AFFINE_PLUS)
    for k in ('G_kwargs', 'D_kwargs'):
        kw = opts[k].get('synthesis_kwargs', opts[k])
        kw['channel_base'], kw['channel_max'] = 512, 32
    kwargs = dict(random_seed=0, G_reg_interval=2, D_reg_interval=2, loss_kwargs=dict(style_mixing_prob=0), **{k: v for k, v in opts.items() if k != 'loss_kwargs'})
    a = training_loop.Trainer(**kwargs)
    a.loss.synthesis_kwargs = dict(noise_mode='const')
    gen = torch.Generator(device=DEV).manual_seed(1)
    batches = [a.synthetic_batch(gen) for _ in range(3)]
    zs = [torch.randn(len(a.phases) * 4, 512, device=DEV, generator=gen) for _ in range(3)]
    for i in range(2):
        a.step(*batches[i], gen_z=zs[i])
    path = str(tmp_path / 'snapshot.pt')
    a.save(path)
    from ga_gan_amd import legacy
    a.save_network_pkl(str(tmp_path / 'network-snapshot.pkl'), training_set_kwargs=dict(path='x.zip'))
    with open(tmp_path / 'network-snapshot.pkl', 'rb') as f:         # the reference's snapshot format, read back
        nets = legacy.load_network_pkl(f)
    assert nets['training_set_kwargs'] == dict(path='x.zip') and nets['augment_pipe'] is None
    for name in ('G', 'D', 'G_ema'):
        for (n, p), (n2, q) in zip(getattr(a, name).state_dict().items(), nets[name].state_dict().items()):
            assert n == n2 and q.device.type == 'cpu' and torch.equal(p.cpu(), q), (name, n)
    b = training_loop.Trainer(**kwargs)
    b.loss.synthesis_kwargs = dict(noise_mode='const')
    b.load(path)
    assert (b.cur_nimg, b.batch_idx) == (a.cur_nimg, a.batch_idx) == (8, 2)
    close(b.loss.pl_mean, a.loss.pl_mean, 1e-7, 'pl_mean')
    for (n, p), q in zip(a.G.named_parameters(), b.G.parameters()):
        assert torch.equal(p, q), n
    for pa, pb in zip(a.phases, b.phases):      # optimizer moments and step counts, parameter by parameter
        for p, q in zip(pa.module.parameters(), pb.module.parameters()):
            sa, sb = pa.opt.state.get(p, {}), pb.opt.state.get(q, {})
            assert sorted(sa) == sorted(sb), pa.name
            for key in sa:
                assert torch.equal(torch.as_tensor(sa[key]).cpu(), torch.as_tensor(sb[key]).cpu()), (pa.name, key)
    a.step(*batches[2], gen_z=zs[2])
    b.step(*batches[2], gen_z=zs[2])             # (the restored trainer keeps training)
    assert (b.cur_nimg, b.batch_idx) == (12, 3)
    assert all(bool(torch.isfinite(p).all()) for p in list(b.G.parameters()) + list(b.D.parameters()))


# ----------------------------------------------------------------------------
# 16-bit runs against the oracle with 16-bit storage emulated (oracle.stylegan2.low_precision): the same layers in
# float32 arithmetic, with every tensor that the product keeps in bfloat16 rounded at the same places.  Layer by
# layer the two agree to 1e-5 .. 4e-4 (float32 sums in a different order move a result across a rounding boundary
# for up to 1 % of the elements); through eight 16-bit layers those differences grow to the size of the rounding
# noise itself, so the emulation cannot track a network element by element.  It gives two things instead:
#   * a second evaluation with the same rounding noise: images, logits and well-conditioned gradients (the
#     discriminator's) agree with it an order of magnitude closer than the float32 fixtures allow;
#   * the size of that noise: for the generator's style gradients (cancelling sums, see above) the emulation's own
#     distance from exact arithmetic bounds what any bfloat16 implementation can reach, and the product is held to
#     1.5 times that distance per tensor.

def _oracle():
    from oracle import stylegan2 as sg
    return sg


def _report(tag, rows):
    print(f'[{tag}] ' + '  '.join(f'{k} {v:.2e}' for k, v in rows))


@pytest.mark.parametrize('case', [c for c in FULL if c.meta['kind'] == 'G'], ids=lambda c: c.meta['tag'])
def test_full_size_generator_bf16_against_rounded_oracle(case):
    from ga_gan_amd.training import training_loop
    sg, m = _oracle(), case.meta
    p = {name: seeded_fill(name, shape) for name, shape in zip(m['names'], m['shapes'])}
    z = torch.randn(1, 512, generator=torch.Generator().manual_seed(m['z_seed']))
    dy = torch.randn(1, 3, m['res'], m['res'], generator=torch.Generator().manual_seed(m['dy_seed']))
    # the oracle, float32 arithmetic, bfloat16 storage in the four highest resolutions
    leaves = {k: (v.clone().requires_grad_(True) if k in m['trainable'] else v) for k, v in p.items()}
    cfg = sg.GeneratorConfig(img_resolution=m['res'], channel_base=m['channel_base'], conv_clamp=256, **_dm(m))
    with sg.low_precision(torch.bfloat16, from_resolution=m['res'] // 8):
        want_img = sg.generator(leaves, z, torch.zeros(1, 0), cfg, noise_mode='const', fused_modconv=False)
        want = torch.autograd.grad((want_img * dy).sum(), [leaves[k] for k in m['trainable']])
    # the product
    G = networks().Generator(
        z_dim=512, c_dim=0, w_dim=512, img_resolution=m['res'], img_channels=3, mapping_kwargs=dict(num_layers=8),
        synthesis_kwargs=dict(channel_base=m['channel_base'], channel_max=512, num_fp16_res=4, conv_clamp=256,
                              generator_requires_grad_parts=m['parts'], **_dm(m)))
    fill_seeded(G, m)
    G.to(DEV).train()
    training_loop.set_requires_grad(G, list(m['parts']))
    img = G(z.to(DEV), torch.zeros(1, 0, device=DEV), noise_mode='const')
    params = dict(G.named_parameters())
    got = torch.autograd.grad((img * dy.to(DEV)).sum(), [params[k] for k in m['trainable']])

    img_err = golden_io.rel_err(img.detach(), want_img.detach())
    num = sum(float((g.detach().double().cpu() - w.double()).square().sum()) for g, w in zip(got, want))
    den = sum(float(w.double().square().sum()) for w in want)
    together = (num / den) ** 0.5
    rows, bad, sq = [], {}, [0.0, 0.0, 0.0]
    for k, g, w in zip(m['trainable'], got, want):
        whole = 'g.' + k in case
        exact = case.t(('xg.' if whole else 'xgs.') + k)        # the same gradient in float64 arithmetic
        pick = (lambda t: t.detach().double().cpu()) if whole else (lambda t: t.detach().double().cpu().flatten()[::61])
        floor, ours = golden_io.rel_err(pick(w), exact), golden_io.rel_err(pick(g), exact)
        sq[0] += float((pick(g) - exact.double()).square().sum())
        sq[1] += float((pick(w) - exact.double()).square().sum())
        sq[2] += float(exact.double().square().sum())
        rows.append((ours / max(floor, 1e-12), ours, floor, k))
        # one tensor: a single draw of the noise on either side.  A one-element tensor (noise_strength: one sum over
        # every pixel of every channel, cancelling to a few percent of its terms) is one draw of one number; its
        # ratio between two draws is unbounded, so it counts in the totals only
        if w.numel() > 1 and ours > max(2e-2, 3 * floor):
            bad[k] = (ours, floor)
    rows.sort(reverse=True)
    ours_all, floor_all = (sq[0] / sq[2]) ** 0.5, (sq[1] / sq[2]) ** 0.5
    _report(m['tag'], [('image vs emulation', img_err), ('all gradients vs emulation', together),
                       ('all gradients vs exact', ours_all), ('the emulation\'s', floor_all),
                       ('worst tensor, ratio to the emulation\'s own error', rows[0][0]), ('its error', rows[0][1])])
    assert img_err < 1e-2
    assert together < 8e-2
    assert ours_all < max(2e-2, 1.5 * floor_all)      # all trainable gradients as one vector: the noise averages out
    assert not bad, bad


@pytest.mark.parametrize('case', [c for c in FULL if c.meta['kind'] == 'D'], ids=lambda c: c.meta['tag'])
def test_full_size_discriminator_bf16_against_rounded_oracle(case):
    """Logits within 1e-2 of their magnitude, every parameter gradient within 2.5e-2 (relative L2) and the image
    gradient within 4e-2 of the emulation (the float32 fixtures above allow 0.1 / 0.25)."""
    sg, m = _oracle(), case.meta
    p = {name: seeded_fill(name, shape) for name, shape in zip(m['names'], m['shapes'])}
    n, res = m['batch'], m['res']
    x = torch.randn(n, 3, res, res, generator=torch.Generator().manual_seed(m['img_seed'])).clamp_(-1, 1)
    names = [k for k in m['names'] if k.endswith('.weight') or k.endswith('.bias')]
    leaves = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in p.items()}
    xo = x.clone().requires_grad_(True)
    cfg = sg.DiscriminatorConfig(img_resolution=res, channel_base=m['channel_base'], conv_clamp=256, mbstd_group_size=m['mbstd_group_size'])
    with sg.low_precision(torch.bfloat16, from_resolution=res // 8):
        want_logits = sg.discriminator(leaves, xo, torch.zeros(n, 0), cfg)
        want = torch.autograd.grad(want_logits.sum(), [xo] + [leaves[k] for k in names])
    D = networks().Discriminator(c_dim=0, img_resolution=res, img_channels=3, channel_base=m['channel_base'], channel_max=512,
                                 num_fp16_res=4, conv_clamp=256, epilogue_kwargs=dict(mbstd_group_size=m['mbstd_group_size']))
    fill_seeded(D, m)
    D.to(DEV).train().requires_grad_(True)
    xg = x.to(DEV).requires_grad_(True)
    logits = D(xg, torch.zeros(n, 0, device=DEV))
    params = dict(D.named_parameters())
    got = torch.autograd.grad(logits.sum(), [xg] + [params[k] for k in names])
    scale = float(want_logits.detach().abs().max())
    logit_err = float((logits.detach().cpu() - want_logits.detach()).abs().max()) / scale
    errs = sorted(((golden_io.rel_err(g, w), k) for k, g, w in zip(['image'] + names, got, want)), reverse=True)
    worst_param = next(e for e in errs if e[1] != 'image')
    image_err = next(e for e in errs if e[1] == 'image')[0]
    _report(m['tag'], [('logits', logit_err), ('image gradient', image_err), ('worst parameter gradient', worst_param[0]),
                       ('median', errs[len(errs) // 2][0])])
    assert logit_err < 1e-2
    assert image_err < 4e-2
    assert worst_param[0] < 2.5e-2, errs[:4]


def test_networks_read_from_a_reference_snapshot_reproduce_its_outputs():
    """`tests/golden/snapshot_ref.pkl` was pickled by the reference (tools/gen_golden_snapshots.py); the networks
    `legacy.load_network_pkl` builds from it give the mapping outputs, images and logits the reference computed with
    the objects it pickled."""
    import json
    import os
    import numpy as np
    from ga_gan_amd import legacy
    with open(os.path.join(golden_io.GOLDEN, 'snapshot_ref.pkl'), 'rb') as f:
        nets = legacy.load_network_pkl(f)
    with np.load(os.path.join(golden_io.GOLDEN, 'snapshot_ref.npz')) as zf:
        want = {k: torch.from_numpy(zf[k]) for k in zf.files if k != 'meta'}
    z = want['z'].to(DEV)
    c = torch.zeros(z.shape[0], 0, device=DEV)
    with torch.no_grad():
        for name in ('G', 'G_ema'):
            G = nets[name].to(DEV)
            close(G.mapping(z, c), want[name + '.ws'], 1e-5, name + '.ws')
            close(G.mapping(z, c, truncation_psi=0.5, truncation_cutoff=4), want[name + '.ws_truncated'], 1e-5, name + '.ws_truncated')
            img = G(z, c, noise_mode='const')
            close(img, want[name + '.img'], 1e-4, name + '.img')
            assert golden_io.rel_err(img, want[name + '.img']) < 1e-3
        img = want['img'].to(DEV)
        close(nets['D'].to(DEV)(img, torch.zeros(img.shape[0], 0, device=DEV)), want['D.logits'], 1e-4, 'D.logits')
