"""The CPU oracle's training objective against the reference's loss phases
(tests/golden/loss.npz: per-phase parameter gradients, w_avg and pl_mean)."""
import pytest
import torch

import golden_io
from oracle import objectives
from oracle import stylegan2 as sg

BUFFER_SUFFIXES = ('resample_filter', 'noise_const', 'w_avg', 'ones_weights', 'offset_mask', 'ones')


def param_names(sd):
    return [k for k in sd if not k.endswith(BUFFER_SUFFIXES)]


@pytest.mark.parametrize('case', golden_io.load('loss'), ids=lambda c: c.meta['tag'])
def test_loss_phases(case):
    m = case.meta
    gp, dp = case.group('G.'), case.group('D.')
    gcfg = sg.GeneratorConfig(img_resolution=m['res'], z_dim=m['z_dim'], w_dim=m['w_dim'], channel_base=m['channel_base'],
                              channel_max=m['channel_max'], conv_clamp=m['conv_clamp'], mapping_layers=m['mapping_layers'],
                              use_domain_modulation=m['parametrization'] is not None,
                              domain_modulation_parametrization=m['parametrization'] or '')
    dcfg = sg.DiscriminatorConfig(img_resolution=m['res'], channel_base=m['channel_base'], channel_max=m['channel_max'],
                                  conv_clamp=m['conv_clamp'], mbstd_group_size=m['mbstd_group_size'])
    obj = objectives.TrainingObjective(gp, gcfg, dp, dcfg, style_mixing_prob=0, r1_gamma=m['r1_gamma'],
                                       pl_weight=m['pl_weight'], pl_decay=m['pl_decay'],
                                       pl_batch_shrink=m['pl_batch_shrink'], noise_mode='const')
    z, c, real = case.t('z'), case.t('c'), case.t('real')
    g_train = objectives.select_trainable(param_names(gp), m['parts'])
    d_train = param_names(dp)
    for phase in ('Gmain', 'Greg', 'Dmain', 'Dreg'):
        params, names = (gp, g_train) if phase[0] == 'G' else (dp, d_train)
        for t in list(gp.values()) + list(dp.values()):
            t.requires_grad_(False)
            t.grad = None
        for n in names:
            params[n].requires_grad_(True)
        assert golden_io.max_err(gp['mapping.w_avg'], case.t(f'{phase}.w_avg_before')) < 1e-6
        obj.accumulate_gradients(phase, real, c, z, c, gain=m['gains'][phase], pl_noise_unit=case.t('pl_noise_unit'))
        want = case.group(f'{phase}.g.')
        assert sorted(want) == sorted(n for n in names if params[n].grad is not None), phase
        for n, g in want.items():
            scale = max(float(g.abs().max()), 1e-3)
            assert golden_io.max_err(params[n].grad, g) <= 2e-4 * scale, (phase, n)
        assert golden_io.max_err(gp['mapping.w_avg'], case.t(f'{phase}.w_avg_after')) < 1e-6
        assert golden_io.max_err(obj.pl_mean, case.t(f'{phase}.pl_mean_after')) < 1e-5 * max(1, float(obj.pl_mean))
