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
            # float atomics in the weight gradients: the order of additions varies from run to run
            assert float((host[n] - replay[n]).abs().max()) <= 1e-3 * scale, (phase.name, n, float((host[n] - replay[n]).abs().max()) / scale)
            checked += 1
    assert checked > 50
