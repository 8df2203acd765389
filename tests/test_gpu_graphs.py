"""Replaying a phase from its recorded HIP graph must yield the gradients that launching it from the host
yields, on the same parameters and inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(use_graphs):
    from ga_gan_amd.training import training_loop
    # tiny learning rates: Adam's first steps are sign-like, and at the default rates a few of them on 8 random
    # images push the toy networks into a regime where two host-launched runs no longer agree either
    # float32 networks: run-to-run noise (the order of float atomics) stays at the 1e-6 level, so the check can be tight
    opts = training_loop.setup_training_options(cfg='paper256', resolution=64, batch_size=8, glrate=1e-5, dlrate=1e-5, fp32=True,
                                                **training_loop.AFFINE_PLUS)
    opts['loss_kwargs'].update(style_mixing_prob=0, pl_weight=0)   # no random draws inside the phases
    trainer = training_loop.Trainer(rank=0, random_seed=3, use_graphs=use_graphs, **opts)
    trainer.loss.synthesis_kwargs = dict(noise_mode='const')
    return trainer


def test_graph_replay_matches_host_launches():
    from ga_gan_amd.training import training_loop
    t = _trainer(True)
    gen = torch.Generator(device='cuda').manual_seed(5)
    c = torch.zeros([8, 0], device='cuda')

    def batch():
        real = torch.randint(0, 256, [8, 3, 64, 64], dtype=torch.uint8, device='cuda', generator=gen)
        return real, torch.randn([len(t.phases) * 8, 512], device='cuda', generator=gen)

    real, z = batch()
    t.step(real, c, gen_z=z)                                   # host launches, then every phase is recorded
    assert set(t._graphs) == {'Gmain', 'Dmain', 'Dreg'}        # (Greg has nothing to do with pl_weight = 0)
    real, z = batch()
    t.step(real, c, gen_z=z)                                   # replays (and trains: parameters now differ from step 0)
    assert t.batch_idx == 2

    real, z = batch()
    real_f = real.to(torch.float32) / 127.5 - 1
    checked = 0
    for i, phase in enumerate(t.phases):
        if phase.name not in t._graphs:
            continue
        g_z = z[8 * i:8 * (i + 1)]
        phase.opt.zero_grad(set_to_none=True)
        training_loop.set_requires_grad(phase.module, parts=t.requires_grad_parts[phase.name])
        t.loss.accumulate_gradients(phase=phase.name, real_img=real_f, real_c=c, gen_z=g_z, gen_c=c, sync=True,
                                    gain=phase.interval)
        phase.module.requires_grad_(False)
        host = {n: p.grad.clone() for n, p in phase.module.named_parameters() if p.grad is not None}
        t._replay_phase(phase, real_f, c, g_z, c)
        replay = {n: p.grad.clone() for n, p in phase.module.named_parameters() if p.grad is not None}
        assert set(host) == set(replay), phase.name
        for n in host:
            scale = max(float(host[n].abs().max()), 1e-6)
            # float atomics in the weight and style gradients: the order of additions varies from run to run, and the
            # style gradients of the 4 x 4 layers are small differences of large sums (seen: 4e-3 of the largest element
            # of synthesis.b4.conv1.affine.weight once in a dozen runs; a replay that read stale memory is off by O(1))
            assert float((host[n] - replay[n]).abs().max()) <= 1e-2 * scale, (phase.name, n, float((host[n] - replay[n]).abs().max()) / scale)
            checked += 1
    assert checked > 50


def test_graph_replay_draws_fresh_style_mixing_and_noise():
    """With the default loss (style mixing 0.9, random per-pixel noise, path-length noise) every draw happens on the
    device inside the recorded phase.  A replay takes the generator state it finds: the same state gives the same
    gradients as before (and the gradients of host launches from that state), another state gives other noise."""
    from ga_gan_amd.training import training_loop
    opts = training_loop.setup_training_options(cfg='paper256', resolution=64, batch_size=8, glrate=1e-5, dlrate=1e-5, fp32=True,
                                                **training_loop.AFFINE_PLUS)
    t = training_loop.Trainer(rank=0, random_seed=3, use_graphs=True, **opts)
    assert t.loss.style_mixing_prob == 0.9 and t.loss.pl_weight > 0 and not t.loss.synthesis_kwargs
    gen = torch.Generator(device='cuda').manual_seed(5)
    c = torch.zeros([8, 0], device='cuda')
    real = torch.randint(0, 256, [8, 3, 64, 64], dtype=torch.uint8, device='cuda', generator=gen)
    z = torch.randn([len(t.phases) * 8, 512], device='cuda', generator=gen)
    t.step(real, c, gen_z=z)
    assert set(t._graphs) == {'Gmain', 'Greg', 'Dmain', 'Dreg'}
    t.step(real, c, gen_z=z)
    real_f = real.to(torch.float32) / 127.5 - 1

    def grads_of(phase, how, seed):
        torch.cuda.manual_seed(seed)
        w_avg, pl_mean = t.G.mapping.w_avg.clone(), t.loss.pl_mean.clone()
        g_z = z[:8]
        if how == 'replay':
            t._replay_phase(phase, real_f, c, g_z, c)
        else:
            phase.opt.zero_grad(set_to_none=True)
            training_loop.set_requires_grad(phase.module, parts=t.requires_grad_parts[phase.name])
            t.loss.accumulate_gradients(phase=phase.name, real_img=real_f, real_c=c, gen_z=g_z, gen_c=c, sync=True, gain=phase.interval)
            phase.module.requires_grad_(False)
        out = {n: p.grad.clone() for n, p in phase.module.named_parameters() if p.grad is not None}
        t.G.mapping.w_avg.copy_(w_avg)
        t.loss.pl_mean.copy_(pl_mean)
        return out

    def distance(a, b):   # all gradients of the phase as one vector, relative L2
        num = sum(float((a[n].double() - b[n].double()).square().sum()) for n in a)
        return (num / sum(float(a[n].double().square().sum()) for n in a)) ** 0.5

    for phase in t.phases:
        first, again, other = grads_of(phase, 'replay', 11), grads_of(phase, 'replay', 11), grads_of(phase, 'replay', 12)
        host = grads_of(phase, 'host', 11)
        assert set(first) == set(host), phase.name
        # run-to-run noise: the order of float atomics, amplified by sums over every pixel that cancel to a small
        # remainder (style and path-length gradients); another draw of the noise moves the gradients by O(1)
        assert distance(first, again) <= 2e-2, (phase.name, distance(first, again))
        assert distance(first, host) <= 2e-2, (phase.name, distance(first, host))
        if phase.name != 'Dreg':                          # (R1 on the real images draws nothing)
            assert distance(first, other) > 0.1, (phase.name, distance(first, other))   # fresh noise per replay


def test_graphs_are_not_recorded_with_an_augment_pipe():
    """The augmentation pipe draws its parameters and padding sizes on the host, which a recording would freeze:
    a trainer with a pipe launches from the host (with a warning), and the adaptive strength keeps moving."""
    from ga_gan_amd.training import training_loop
    opts = training_loop.setup_training_options(cfg='paper256', resolution=64, batch_size=8, glrate=1e-5, dlrate=1e-5,
                                                **training_loop.AFFINE_PLUS)
    pipe = training_loop.make_augment_pipe('bgc', 0.2, torch.device('cuda'))
    with pytest.warns(UserWarning, match='use_graphs is ignored'):
        t = training_loop.Trainer(rank=0, random_seed=3, use_graphs=True, augment_pipe=pipe, ada_target=0.6, ada_interval=2, ada_kimg=1, **opts)
    assert t.use_graphs is False and t.loss.grad_sync is t.reducer
    real, real_c = t.synthetic_batch(torch.Generator(device='cuda').manual_seed(1))
    p0 = float(pipe.p)
    for _ in range(3):                                    # (one adjustment of the strength, after the second step)
        t.step(real, real_c)
    assert not t._graphs
    assert abs(abs(float(pipe.p) - p0) - 8 * 2 / 1000) < 1e-6
