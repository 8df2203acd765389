"""Oracle (test infrastructure): the StyleGAN2 training objective on CPU.

Restates `StyleGAN2Loss.accumulate_gradients` (DissimilarDomains/training/loss.py:69-152):
non-saturating logistic loss, lazy R1 on reals (loss.py:127-152), path-length
regularisation on a shrunk batch (loss.py:89-111) and style mixing
(loss.py:44-60), over the functional networks of `oracle.stylegan2`.
Gradients accumulate into the `.grad` of whichever parameter tensors have
`requires_grad` set, exactly like the reference's `.backward()` calls.
"""
import math

import torch
import torch.nn.functional as F

from . import stylegan2 as sg


class TrainingObjective:
    def __init__(self, g_params, g_cfg, d_params, d_cfg, style_mixing_prob=0.9, r1_gamma=10.0, pl_batch_shrink=2,
                 pl_decay=0.01, pl_weight=2.0, noise_mode='random', training=True):
        self.gp, self.gcfg, self.dp, self.dcfg = g_params, g_cfg, d_params, d_cfg
        self.style_mixing_prob = style_mixing_prob
        self.r1_gamma = r1_gamma
        self.pl_batch_shrink = pl_batch_shrink
        self.pl_decay = pl_decay
        self.pl_weight = pl_weight
        self.noise_mode = noise_mode
        self.training = training
        self.pl_mean = torch.zeros([])
        self.stats = {}

    def _map(self, z, c, update_w_avg=True):
        ws = sg.mapping(self.gp, z, c, self.gcfg)
        beta = self.gcfg.w_avg_beta
        if self.training and update_w_avg and beta is not None:
            with torch.no_grad():
                self.gp['mapping.w_avg'].copy_(sg.w_avg_update(self.gp['mapping.w_avg'], ws[:, 0], beta))
        return ws

    def run_G(self, z, c, set_w_requires_grad=False):
        ws = self._map(z, c)
        if self.style_mixing_prob > 0:
            cutoff = torch.empty([], dtype=torch.int64).random_(1, ws.shape[1])
            cutoff = torch.where(torch.rand([]) < self.style_mixing_prob, cutoff, torch.full_like(cutoff, ws.shape[1]))
            ws2 = self._map(torch.randn_like(z), c, update_w_avg=False)
            keep = (torch.arange(ws.shape[1]) < cutoff).reshape(1, -1, 1)
            ws = torch.where(keep, ws, ws2)
        if set_w_requires_grad:
            ws.requires_grad_(True)
        img = sg.synthesis(self.gp, ws, self.gcfg, noise_mode=self.noise_mode, fused_modconv=False)
        return img, ws

    def run_D(self, img, c):
        return sg.discriminator(self.dp, img, c, self.dcfg)

    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, gain, pl_noise_unit=None):
        assert phase in ('Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth')
        do_Gmain = phase in ('Gmain', 'Gboth')
        do_Dmain = phase in ('Dmain', 'Dboth')
        do_Gpl = phase in ('Greg', 'Gboth') and self.pl_weight != 0
        do_Dr1 = phase in ('Dreg', 'Dboth') and self.r1_gamma != 0

        if do_Gmain:
            gen_img, _ = self.run_G(gen_z, gen_c)
            gen_logits = self.run_D(gen_img, gen_c)
            loss_Gmain = F.softplus(-gen_logits)
            self.stats['Loss/G/loss'] = loss_Gmain.detach()
            loss_Gmain.mean().mul(gain).backward()

        if do_Gpl:
            n = gen_z.shape[0] // self.pl_batch_shrink
            gen_img, gen_ws = self.run_G(gen_z[:n], gen_c[:n], set_w_requires_grad=True)
            unit = torch.randn_like(gen_img) if pl_noise_unit is None else pl_noise_unit
            pl_noise = unit / math.sqrt(gen_img.shape[2] * gen_img.shape[3])
            pl_grads, = torch.autograd.grad([(gen_img * pl_noise).sum()], [gen_ws], create_graph=True)
            pl_lengths = pl_grads.square().sum(2).mean(1).sqrt()
            pl_mean = self.pl_mean.lerp(pl_lengths.mean(), self.pl_decay)
            self.pl_mean = pl_mean.detach().clone()
            pl_penalty = (pl_lengths - pl_mean).square()
            self.stats['Loss/pl_penalty'] = pl_penalty.detach()
            loss_Gpl = pl_penalty * self.pl_weight
            (gen_img[:, 0, 0, 0] * 0 + loss_Gpl).mean().mul(gain).backward()

        loss_Dgen = 0
        if do_Dmain:
            gen_img, _ = self.run_G(gen_z, gen_c)
            gen_logits = self.run_D(gen_img, gen_c)
            loss_Dgen = F.softplus(gen_logits)
            loss_Dgen.mean().mul(gain).backward()

        if do_Dmain or do_Dr1:
            real_tmp = real_img.detach().requires_grad_(do_Dr1)
            real_logits = self.run_D(real_tmp, real_c)
            self.stats['Loss/signs/real'] = real_logits.detach().sign()
            loss_Dreal = 0
            if do_Dmain:
                loss_Dreal = F.softplus(-real_logits)
                self.stats['Loss/D/loss'] = (loss_Dgen + loss_Dreal).detach()
            loss_Dr1 = 0
            if do_Dr1:
                r1_grads, = torch.autograd.grad([real_logits.sum()], [real_tmp], create_graph=True)
                r1_penalty = r1_grads.square().sum([1, 2, 3])
                loss_Dr1 = r1_penalty * (self.r1_gamma / 2)
                self.stats['Loss/r1_penalty'] = r1_penalty.detach()
            (real_logits * 0 + loss_Dreal + loss_Dr1).mean().mul(gain).backward()


# Parameter-group selection for few-shot training: which generator tensors a
# `--generator-requires-grad-parts` entry names (training/training_loop.py:41-95).

def _in_block(name, res):
    return ('synthesis' in name) if res is None else (f'synthesis.b{res}' in name)


PART_FILTERS = {
    'mapping': lambda res, n: 'mapping' in n,
    'tRGB_affine': lambda res, n: _in_block(n, res) and 'torgb.affine' in n,
    'tRGB_conv': lambda res, n: _in_block(n, res) and ('torgb.weight' in n or 'torgb.bias' in n)
    and 'affine' not in n and 'offset' not in n,
    'tRGB_offset': lambda res, n: _in_block(n, res) and 'torgb.offset' in n and 'torgb.weights_offset' not in n,
    'tRGB_weights_offset': lambda res, n: _in_block(n, res) and 'torgb.weights_offset' in n,
    'tRGB_affine_weights_offset': lambda res, n: _in_block(n, res) and 'torgb.affine.weights_offset' in n,
    'synt_affine': lambda res, n: _in_block(n, res) and 'conv' in n and 'affine' in n,
    'synt_conv': lambda res, n: _in_block(n, res) and 'conv' in n
    and ('weight' in n or 'noise_strength' in n or 'bias' in n) and 'affine' not in n and 'offset' not in n,
    'synt_const': lambda res, n: _in_block(n, res) and 'const' in n,
    'synt_offset': lambda res, n: _in_block(n, res) and 'conv' in n and 'offset' in n and 'weights_offset' not in n,
    'synt_weights_offset': lambda res, n: _in_block(n, res) and 'conv' in n and 'affine' not in n
    and 'weights_offset' in n,
    'synt_affine_weights_offset': lambda res, n: _in_block(n, res) and 'conv' in n and 'affine.weights_offset' in n,
}


def select_trainable(param_names, parts):
    """Names from `param_names` switched on by `parts` ('all', '<group>' or '<group>.b<res>')."""
    if 'all' in parts:
        return list(param_names)
    chosen = set()
    for group, flt in PART_FILTERS.items():
        for res in (1024, 512, 256, 128, 64, 32, 16, 8, 4, None):
            target = group if res is None else f'{group}.b{res}'
            if target in parts:
                chosen.update(n for n in param_names if flt(res, n))
    return [n for n in param_names if n in chosen]
