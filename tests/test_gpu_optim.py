"""GPU parity of the fused optimizer step (sg_adam_step) and the moving average (sg_lerp_multi)
against the CPU oracle, through the C ABI binding.  float32 arithmetic on both sides: tolerance 2e-6
relative (the device's division / square root round like the host's; the fused multiply-adds the
compiler forms do not)."""
import numpy as np
import pytest
import torch

from oracle import optim as oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_adam_step_matches_oracle_over_steps():
    from ga_gan_amd import optim_plugin
    rng = np.random.default_rng(0)
    shapes = [(7,), (3, 5), (64, 32, 3, 3), (1,), (4097,), (512, 512)] + [(3,)] * 70    # 76 tensors: two launches
    params = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    ms = [np.zeros_like(p) for p in params]
    vs = [np.zeros_like(p) for p in params]
    tparams = [torch.nn.Parameter(torch.tensor(p, device=DEV)) for p in params]
    opt = optim_plugin.FusedAdam(tparams, lr=0.0025, betas=(0.0, 0.99 ** (16 / 17)), eps=1e-8)
    for step in range(1, 5):
        grads = [(rng.standard_normal(s) * 10.0 ** rng.integers(-4, 3)).astype(np.float32) for s in shapes]
        grads[0][1] = np.nan
        grads[1][0, 2] = np.inf
        grads[2][1, 0, 2, 1] = -np.inf
        skipped = {3} if step == 2 else set()          # a parameter without a gradient keeps its state and step count
        for i, (tp, g) in enumerate(zip(tparams, grads)):
            tp.grad = None if i in skipped else torch.tensor(g, device=DEV)
        opt.step()
        for i in range(len(params)):
            if i in skipped:
                continue
            n_updates = step - (1 if (i == 3 and step > 2) else 0)
            params[i], g_clean, ms[i], vs[i] = oracle.adam_step(params[i], grads[i], ms[i], vs[i], n_updates, 0.0025, 0.0,
                                                                0.99 ** (16 / 17), 1e-8)
            np.testing.assert_allclose(tparams[i].detach().cpu().numpy(), params[i], rtol=2e-6, atol=1e-7)
            np.testing.assert_array_equal(tparams[i].grad.cpu().numpy(), g_clean)
            np.testing.assert_allclose(opt.state[tparams[i]]['exp_avg_sq'].cpu().numpy(), vs[i], rtol=2e-6, atol=1e-30)
            np.testing.assert_array_equal(opt.state[tparams[i]]['exp_avg'].cpu().numpy(), ms[i])   # beta1 = 0: the gradient itself


def test_state_dict_is_interchangeable_with_torch_adam():
    from ga_gan_amd import optim_plugin
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(33, 5, device=DEV)), torch.nn.Parameter(torch.randn(9, device=DEV))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    fused = optim_plugin.FusedAdam(a, lr=0.01, betas=(0.5, 0.9), eps=1e-8)
    ref = torch.optim.Adam(b, lr=0.01, betas=(0.5, 0.9), eps=1e-8)
    for _ in range(3):
        for p, q in zip(a, b):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        fused.step()
        ref.step()
    for p, q in zip(a, b):
        torch.testing.assert_close(p, q, rtol=2e-6, atol=1e-7)
    ref2 = torch.optim.Adam(b, lr=0.01, betas=(0.5, 0.9), eps=1e-8)
    ref2.load_state_dict(fused.state_dict())           # same layout: step / exp_avg / exp_avg_sq per parameter
    assert int(ref2.state[b[0]]['step']) == 3


def test_ema_matches_oracle():
    from ga_gan_amd import optim_plugin
    rng = np.random.default_rng(2)
    shapes = [(5,), (130, 7), (4096 * 3 + 1,)]
    e = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    p = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    te = [torch.tensor(v, device=DEV) for v in e]
    tp = [torch.tensor(v, device=DEV) for v in p]
    optim_plugin.ema_(te, tp, 0.998)
    for got, ee, pp in zip(te, e, p):
        np.testing.assert_allclose(got.cpu().numpy(), oracle.ema(ee, pp, 0.998), rtol=2e-6, atol=1e-7)


def test_rejects_bad_tensors():
    from ga_gan_amd import optim_plugin
    p = torch.zeros(4, device=DEV)
    with pytest.raises(RuntimeError):
        optim_plugin.adam_step([p], [torch.zeros(4, device=DEV, dtype=torch.bfloat16)], [p.clone()], [p.clone()], [1], 1e-3, 0.9, 0.99, 1e-8)
    with pytest.raises(RuntimeError):
        optim_plugin.adam_step([p], [torch.zeros(4)], [p.clone()], [p.clone()], [1], 1e-3, 0.9, 0.99, 1e-8)
    with pytest.raises(RuntimeError):   # step counts from 1
        optim_plugin.adam_step([p], [p.clone()], [p.clone()], [p.clone()], [0], 1e-3, 0.9, 0.99, 1e-8)


def test_updates_advance_tensor_versions():
    """Caches keyed on `Tensor._version` (packed weights of frozen networks) must see the kernel's writes."""
    from ga_gan_amd import optim_plugin
    p = torch.nn.Parameter(torch.randn(10, device=DEV))
    p.grad = torch.randn(10, device=DEV)
    opt = optim_plugin.FusedAdam([p], lr=0.01)
    before = p._version
    opt.step()
    assert p._version > before
    e = torch.zeros(10, device=DEV)
    before = e._version
    optim_plugin.ema_([e], [p.detach()], 0.9)
    assert e._version > before


def test_loaded_step_counts_stay_on_the_host(tmp_path):
    """A snapshot read with map_location=<device> carries the step counts as device tensors; after load_state_dict
    they live on the host again, so a step issues no device-to-host read (checked with torch's sync debug mode)."""
    from ga_gan_amd import optim_plugin
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(17, 3, device=DEV)), torch.nn.Parameter(torch.randn(5, device=DEV))]
    opt = optim_plugin.FusedAdam(params, lr=0.01, betas=(0.5, 0.9))
    for p in params:
        p.grad = torch.randn_like(p)
    opt.step()
    torch.save(opt.state_dict(), tmp_path / 'opt.pt')
    loaded = torch.load(tmp_path / 'opt.pt', map_location=DEV, weights_only=True)
    assert loaded['state'][0]['step'].is_cuda
    fresh = optim_plugin.FusedAdam(params, lr=0.01, betas=(0.5, 0.9))
    fresh.load_state_dict(loaded)
    assert all(fresh.state[p]['step'].device.type == 'cpu' and int(fresh.state[p]['step']) == 1 for p in params)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        fresh.step()
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert all(int(fresh.state[p]['step']) == 2 for p in params)
