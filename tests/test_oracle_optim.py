"""The optimizer oracle against the library the reference calls (torch.optim.Adam on CPU, after
`nan_to_num` of the gradients), over several steps with NaN / Inf gradients and the
lazy-regularisation betas of the reference's loop."""
import numpy as np
import pytest
import torch

from oracle import optim as oracle


@pytest.mark.parametrize('betas', [(0.0, 0.99), (0.0, 0.99 ** (16 / 17)), (0.9, 0.999)])
def test_adam_oracle_matches_torch(betas):
    rng = np.random.default_rng(0)
    shapes = [(7,), (3, 5), (2, 3, 3, 3), (1,)]
    params = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    tparams = [torch.nn.Parameter(torch.tensor(p)) for p in params]
    opt = torch.optim.Adam(tparams, lr=0.0025, betas=betas, eps=1e-8)
    ms = [np.zeros_like(p) for p in params]
    vs = [np.zeros_like(p) for p in params]
    for step in range(1, 6):
        grads = [(rng.standard_normal(s) * 10.0 ** rng.integers(-4, 3)).astype(np.float32) for s in shapes]
        grads[0][1] = np.nan
        grads[1][0, 2] = np.inf
        grads[2][1, 0, 2, 1] = -np.inf
        for tp, g in zip(tparams, grads):
            tp.grad = torch.nan_to_num(torch.tensor(g), nan=0.0, posinf=1e5, neginf=-1e5)
        opt.step()
        for i in range(len(params)):
            params[i], _, ms[i], vs[i] = oracle.adam_step(params[i], grads[i], ms[i], vs[i], step, 0.0025, betas[0], betas[1], 1e-8)
            np.testing.assert_allclose(params[i], tparams[i].detach().numpy(), rtol=2e-6, atol=1e-7)
            np.testing.assert_allclose(ms[i], opt.state[tparams[i]]['exp_avg'].numpy(), rtol=2e-6, atol=1e-30)
            np.testing.assert_allclose(vs[i], opt.state[tparams[i]]['exp_avg_sq'].numpy(), rtol=2e-6, atol=1e-30)


def test_ema_oracle_matches_torch():
    rng = np.random.default_rng(1)
    p, e = rng.standard_normal(100).astype(np.float32), rng.standard_normal(100).astype(np.float32)
    want = torch.tensor(p).lerp(torch.tensor(e), 0.998).numpy()
    np.testing.assert_allclose(oracle.ema(e, p, 0.998), want, rtol=1e-6, atol=1e-7)
