"""Training objective of the few-shot loop: non-saturating logistic loss, lazy R1 on
real images and path-length regularisation, with the reference's phase protocol
(`accumulate_gradients(phase, real_img, real_c, gen_z, gen_c, sync, gain)`,
training/loss.py:69-152).  Gradient synchronisation across GPUs: where the reference enters `ddp_sync(module, sync)`
around the forward pass whose backward is the phase's last one, this loss arms `grad_sync`
(the trainer's GradReducer) right before that backward call, so that gradients are exchanged
while the rest of the backward pass still runs."""
import contextlib

import numpy as np
import torch

from ..torch_utils import training_stats
from ..torch_utils.ops import conv2d_gradfix
from ..torch_utils.ops import fused_layer


def _interleave(a, b, groups):
    """[n, ...] x 2 -> [2n, ...]: rows split into `groups` runs; run g of a, then run g of b, for g = 0, 1, ..."""
    run = a.shape[0] // groups
    return torch.cat([a.unflatten(0, (groups, run)), b.unflatten(0, (groups, run))], dim=1).flatten(0, 1)


class Loss:
    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync, gain):
        raise NotImplementedError()


class StyleGAN2Loss(Loss):
    def __init__(self, device, G_mapping, G_synthesis, D, augment_pipe=None, style_mixing_prob=0.9, r1_gamma=10,
                 pl_batch_shrink=2, pl_decay=0.01, pl_weight=2):
        super().__init__()
        self.device = device
        self.G_mapping = G_mapping
        self.G_synthesis = G_synthesis
        self.D = D
        self.augment_pipe = augment_pipe
        self.style_mixing_prob = style_mixing_prob
        self.r1_gamma = r1_gamma
        self.pl_batch_shrink = pl_batch_shrink
        self.pl_decay = pl_decay
        self.pl_weight = pl_weight
        self.pl_mean = torch.zeros([], device=device)
        self.synthesis_kwargs = {}      # e.g. dict(noise_mode='const') for deterministic runs
        self.grad_sync = None           # object with arm(): told when the next backward call's gradients are final
        self.merge_D_passes = True      # Dmain: generated and real half in one discriminator pass (same result)
        self._mbstd_sizes = None

    def _arm(self, sync):
        if sync and self.grad_sync is not None:
            self.grad_sync.arm()

    def _merged_groups(self, n_gen, n_real):
        """The discriminator's minibatch-std group size when the generated and the real half of Dmain can share one
        discriminator pass (it is the number of runs `_interleave` cuts each half into), else None.  The reference runs two passes (loss.py:118-125 and :129-151); samples only
        meet in the minibatch standard deviation, whose groups are the samples with equal index modulo
        batch / group_size (networks.py:1291-1293), so one pass over [gen rows of group-row 0, real rows of group-row 0,
        gen rows of group-row 1, ...] forms exactly the groups of the two separate passes and every logit is the same."""
        if not self.merge_D_passes or n_gen != n_real or n_gen == 0:
            return None
        if self._mbstd_sizes is None:
            known = type(self.D).__module__.endswith('training.networks') and type(self.D).__name__ == 'Discriminator'
            layers = [m for m in self.D.modules() if type(m).__name__ == 'MinibatchStdLayer']
            self._mbstd_sizes = [m.group_size for m in layers] if known else [None]
        if self._mbstd_sizes != [None] and (n_gen + n_real) * self.D.largest_activation_elements() > 2 ** 31 - 1:
            return None              # one pass would make a tensor beyond the kernels' 32-bit indexing: two passes
        group = None
        for size in self._mbstd_sizes:
            if size is None or size > n_gen or n_gen % size != 0 or (group is not None and size != group):
                return None          # whole-batch groups (or a discriminator this loss does not know): two passes
            group = size
        return group if group is not None else 1

    def run_G(self, z, c, sync, set_w_requires_grad=False):
        ws = self.G_mapping(z, c)
        if self.style_mixing_prob > 0:
            with torch.autograd.profiler.record_function('style_mixing'):
                cutoff = torch.empty([], dtype=torch.int64, device=ws.device).random_(1, ws.shape[1])
                cutoff = torch.where(torch.rand([], device=ws.device) < self.style_mixing_prob, cutoff,
                                     torch.full_like(cutoff, ws.shape[1]))
                # the cutoff stays on the device: blend with a mask instead of slicing at a host-side index
                keep = (torch.arange(ws.shape[1], device=ws.device) < cutoff).reshape(1, -1, 1)
                ws = torch.where(keep, ws, self.G_mapping(torch.randn_like(z), c, skip_w_avg_update=True))
        if set_w_requires_grad:
            ws.requires_grad_(True)
        img = self.G_synthesis(ws, **self.synthesis_kwargs)
        return img, ws

    def run_D(self, img, c, sync):
        if self.augment_pipe is not None:
            img = self.augment_pipe(img)
        return self.D(img, c)

    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync, gain, pl_noise_unit=None):
        assert phase in ['Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth']
        do_Gmain = (phase in ['Gmain', 'Gboth'])
        do_Dmain = (phase in ['Dmain', 'Dboth'])
        do_Gpl = (phase in ['Greg', 'Gboth']) and (self.pl_weight != 0)
        do_Dr1 = (phase in ['Dreg', 'Dboth']) and (self.r1_gamma != 0)

        # Gmain: maximise logits of generated images.
        if do_Gmain:
            with torch.autograd.profiler.record_function('Gmain_forward'):
                gen_img, _ = self.run_G(gen_z, gen_c, sync=(sync and not do_Gpl))
                gen_logits = self.run_D(gen_img, gen_c, sync=False)
                training_stats.report('Loss/scores/fake', gen_logits)
                training_stats.report('Loss/signs/fake', gen_logits.sign())
                loss_Gmain = torch.nn.functional.softplus(-gen_logits)
                training_stats.report('Loss/G/loss', loss_Gmain)
            with torch.autograd.profiler.record_function('Gmain_backward'):
                self._arm(sync and not do_Gpl)   # (Gboth: the path-length pass adds to the same gradients)
                loss_Gmain.mean().mul(gain).backward()

        # Gpl: path-length regularisation on a shrunk batch.
        if do_Gpl:
            with torch.autograd.profiler.record_function('Gpl_forward'):
                batch_size = gen_z.shape[0] // self.pl_batch_shrink
                with fused_layer.differentiable_twice():
                    gen_img, gen_ws = self.run_G(gen_z[:batch_size], gen_c[:batch_size], sync=sync, set_w_requires_grad=True)
                unit = torch.randn_like(gen_img) if pl_noise_unit is None else pl_noise_unit
                pl_noise = unit / np.sqrt(gen_img.shape[2] * gen_img.shape[3])
                with torch.autograd.profiler.record_function('pl_grads'), conv2d_gradfix.no_weight_gradients():
                    pl_grads = torch.autograd.grad(outputs=[(gen_img * pl_noise).sum()], inputs=[gen_ws],
                                                   create_graph=True, only_inputs=True, allow_unused=True)[0]
                pl_lengths = pl_grads.square().sum(2).mean(1).sqrt()
                pl_mean = self.pl_mean.lerp(pl_lengths.mean(), self.pl_decay)
                self.pl_mean.copy_(pl_mean.detach())
                pl_penalty = (pl_lengths - pl_mean).square()
                training_stats.report('Loss/pl_penalty', pl_penalty)
                loss_Gpl = pl_penalty * self.pl_weight
                training_stats.report('Loss/G/reg', loss_Gpl)
            with torch.autograd.profiler.record_function('Gpl_backward'):
                self._arm(sync)
                (gen_img[:, 0, 0, 0] * 0 + loss_Gpl).mean().mul(gain).backward()

        # Dmain, both halves in one discriminator pass where that is the same computation.
        groups = self._merged_groups(gen_z.shape[0], real_img.shape[0]) if do_Dmain and not do_Dr1 else None
        if groups is not None:
            with torch.autograd.profiler.record_function('Dmain_forward'):
                gen_img, _ = self.run_G(gen_z, gen_c, sync=False)
                if self.augment_pipe is not None:
                    gen_img, real_aug = self.augment_pipe(gen_img), self.augment_pipe(real_img.detach())
                else:
                    real_aug = real_img.detach()
                logits = self.D(_interleave(gen_img, real_aug, groups), _interleave(gen_c, real_c, groups))
                per_group = logits.unflatten(0, (groups, logits.shape[0] // groups))
                half = per_group.shape[1] // 2
                gen_logits, real_logits = per_group[:, :half].flatten(0, 1), per_group[:, half:].flatten(0, 1)
                training_stats.report('Loss/scores/fake', gen_logits)
                training_stats.report('Loss/signs/fake', gen_logits.sign())
                training_stats.report('Loss/scores/real', real_logits)
                training_stats.report('Loss/signs/real', real_logits.sign())
                loss_Dgen = torch.nn.functional.softplus(gen_logits)
                loss_Dreal = torch.nn.functional.softplus(-real_logits)
                training_stats.report('Loss/D/loss', loss_Dgen + loss_Dreal)
            with torch.autograd.profiler.record_function('Dmain_backward'):
                self._arm(sync)
                (loss_Dgen.mean() + loss_Dreal.mean()).mul(gain).backward()
            return

        # Dmain: minimise logits of generated images.
        loss_Dgen = 0
        if do_Dmain:
            with torch.autograd.profiler.record_function('Dgen_forward'):
                gen_img, _ = self.run_G(gen_z, gen_c, sync=False)
                gen_logits = self.run_D(gen_img, gen_c, sync=False)
                training_stats.report('Loss/scores/fake', gen_logits)
                training_stats.report('Loss/signs/fake', gen_logits.sign())
                loss_Dgen = torch.nn.functional.softplus(gen_logits)
            with torch.autograd.profiler.record_function('Dgen_backward'):
                loss_Dgen.mean().mul(gain).backward()

        # Dmain: maximise logits of real images.  Dr1: gradient penalty on reals.
        if do_Dmain or do_Dr1:
            name = 'Dreal_Dr1' if do_Dmain and do_Dr1 else 'Dreal' if do_Dmain else 'Dr1'
            with torch.autograd.profiler.record_function(name + '_forward'):
                real_img_tmp = real_img.detach().requires_grad_(do_Dr1)
                with fused_layer.differentiable_twice() if do_Dr1 else contextlib.nullcontext():
                    real_logits = self.run_D(real_img_tmp, real_c, sync=sync)
                training_stats.report('Loss/scores/real', real_logits)
                training_stats.report('Loss/signs/real', real_logits.sign())
                loss_Dreal = 0
                if do_Dmain:
                    loss_Dreal = torch.nn.functional.softplus(-real_logits)
                    training_stats.report('Loss/D/loss', loss_Dgen + loss_Dreal)
                loss_Dr1 = 0
                if do_Dr1:
                    with torch.autograd.profiler.record_function('r1_grads'), conv2d_gradfix.no_weight_gradients():
                        r1_grads = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_img_tmp],
                                                       create_graph=True, only_inputs=True)[0]
                    r1_penalty = r1_grads.square().sum([1, 2, 3])
                    loss_Dr1 = r1_penalty * (self.r1_gamma / 2)
                    training_stats.report('Loss/r1_penalty', r1_penalty)
                    training_stats.report('Loss/D/reg', loss_Dr1)
            with torch.autograd.profiler.record_function(name + '_backward'):
                self._arm(sync)                  # (the Dgen pass before it is not final)
                (real_logits * 0 + loss_Dreal + loss_Dr1).mean().mul(gain).backward()
