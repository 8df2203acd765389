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

    # ---- building blocks of the phases ------------------------------------------------------------------------
    # Every phase is: a forward pass inside a '<term>_forward' profiler scope that ends in a per-sample loss, then one
    # backward call of mean(loss) * gain inside '<term>_backward'.  The reducer is armed in front of the backward call
    # that completes the phase's gradients, and only there.

    @staticmethod
    def _scope(term, part):
        return torch.autograd.profiler.record_function(f'{term}_{part}')

    @staticmethod
    def _report_scores(kind, logits):
        training_stats.report(f'Loss/scores/{kind}', logits)
        training_stats.report(f'Loss/signs/{kind}', logits.sign())

    def _backward(self, term, per_sample, gain, final, anchor=None):
        """mean(per_sample) * gain -> gradients.  `anchor` (a tensor of the forward pass, one element per sample) is
        added with weight zero: every parameter behind it then receives a gradient tensor - zeros where the loss
        does not reach it - which is what the reference's `x * 0 + loss` idiom (loss.py:111, :152) produces, and
        what decides which parameters the optimizer steps."""
        with self._scope(term, 'backward'):
            if final:
                self._arm(True)
            if anchor is not None:
                per_sample = anchor * 0 + per_sample
            per_sample.mean().mul(gain).backward()

    def _g_main(self, gen_z, gen_c, gain, final):
        """Generator: non-saturating loss softplus(-D(G(z)))  (reference loss.py:77-86)."""
        with self._scope('Gmain', 'forward'):
            fake, _ = self.run_G(gen_z, gen_c, sync=final)
            logits = self.run_D(fake, gen_c, sync=False)
            self._report_scores('fake', logits)
            loss = torch.nn.functional.softplus(-logits)
            training_stats.report('Loss/G/loss', loss)
        self._backward('Gmain', loss, gain, final)

    def _g_path_length(self, gen_z, gen_c, gain, final, noise_unit):
        """Generator: path-length penalty on the first batch / pl_batch_shrink samples  (reference loss.py:89-111)."""
        with self._scope('Gpl', 'forward'):
            count = gen_z.shape[0] // self.pl_batch_shrink
            with fused_layer.differentiable_twice():
                fake, ws = self.run_G(gen_z[:count], gen_c[:count], sync=final, set_w_requires_grad=True)
            direction = (torch.randn_like(fake) if noise_unit is None else noise_unit) / np.sqrt(fake.shape[2] * fake.shape[3])
            with torch.autograd.profiler.record_function('pl_grads'), conv2d_gradfix.no_weight_gradients(all_parameters=True):
                jvp, = torch.autograd.grad(outputs=[(fake * direction).sum()], inputs=[ws], create_graph=True,
                                           only_inputs=True, allow_unused=True)
            lengths = jvp.square().sum(2).mean(1).sqrt()
            mean = self.pl_mean.lerp(lengths.mean(), self.pl_decay)
            self.pl_mean.copy_(mean.detach())
            penalty = (lengths - mean).square()
            training_stats.report('Loss/pl_penalty', penalty)
            loss = penalty * self.pl_weight
            training_stats.report('Loss/G/reg', loss)
        self._backward('Gpl', loss, gain, final, anchor=fake[:, 0, 0, 0])

    def _d_main_merged(self, groups, real_img, real_c, gen_z, gen_c, gain, final):
        """Discriminator: softplus(D(G(z))) + softplus(-D(real)) from ONE pass over both halves (see _merged_groups)."""
        with self._scope('Dmain', 'forward'):
            fake, _ = self.run_G(gen_z, gen_c, sync=False)
            real = real_img.detach()
            if self.augment_pipe is not None:
                fake, real = self.augment_pipe(fake), self.augment_pipe(real)
            logits = self.D(_interleave(fake, real, groups), _interleave(gen_c, real_c, groups))
            runs = logits.unflatten(0, (groups, logits.shape[0] // groups))      # [group row, fake run | real run, 1]
            half = runs.shape[1] // 2
            fake_logits, real_logits = runs[:, :half].flatten(0, 1), runs[:, half:].flatten(0, 1)
            self._report_scores('fake', fake_logits)
            self._report_scores('real', real_logits)
            loss_fake = torch.nn.functional.softplus(fake_logits)
            loss_real = torch.nn.functional.softplus(-real_logits)
            training_stats.report('Loss/D/loss', loss_fake + loss_real)
        with self._scope('Dmain', 'backward'):
            if final:
                self._arm(True)
            (loss_fake.mean() + loss_real.mean()).mul(gain).backward()

    def _d_fake(self, gen_z, gen_c, gain):
        """Discriminator on generated images: softplus(D(G(z)))  (reference loss.py:115-123).  Never the last
        backward call of its phase: the real half follows."""
        with self._scope('Dgen', 'forward'):
            fake, _ = self.run_G(gen_z, gen_c, sync=False)
            logits = self.run_D(fake, gen_c, sync=False)
            self._report_scores('fake', logits)
            loss = torch.nn.functional.softplus(logits)
        self._backward('Dgen', loss, gain, final=False)
        return loss

    def _d_real(self, real_img, real_c, gain, final, with_main, with_r1, loss_fake):
        """Discriminator on real images: softplus(-D(x)) and / or the R1 penalty gamma / 2 * |dD/dx|^2
        (reference loss.py:127-152)."""
        term = 'Dreal_Dr1' if with_main and with_r1 else 'Dreal' if with_main else 'Dr1'
        with self._scope(term, 'forward'):
            x = real_img.detach().requires_grad_(with_r1)
            with fused_layer.differentiable_twice() if with_r1 else contextlib.nullcontext():
                logits = self.run_D(x, real_c, sync=final)
            self._report_scores('real', logits)
            total = 0
            if with_main:
                main = torch.nn.functional.softplus(-logits)
                training_stats.report('Loss/D/loss', loss_fake + main)
                total = total + main
            if with_r1:
                with torch.autograd.profiler.record_function('r1_grads'), conv2d_gradfix.no_weight_gradients(all_parameters=True):
                    slope, = torch.autograd.grad(outputs=[logits.sum()], inputs=[x], create_graph=True, only_inputs=True)
                penalty = slope.square().sum([1, 2, 3])
                reg = penalty * (self.r1_gamma / 2)
                training_stats.report('Loss/r1_penalty', penalty)
                training_stats.report('Loss/D/reg', reg)
                total = total + reg
        self._backward(term, total, gain, final, anchor=logits)

    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync, gain, pl_noise_unit=None):
        """Gradients of one phase into the .grad of the parameters that require them.  `sync`: this is the last
        accumulation round of the phase, i.e. the gradients are final after it."""
        if phase not in ('Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth'):
            raise ValueError(f'unknown phase {phase!r}')
        g_main = phase in ('Gmain', 'Gboth')
        g_reg = phase in ('Greg', 'Gboth') and self.pl_weight != 0
        d_main = phase in ('Dmain', 'Dboth')
        d_reg = phase in ('Dreg', 'Dboth') and self.r1_gamma != 0

        if g_main:
            self._g_main(gen_z, gen_c, gain, final=sync and not g_reg)   # (Gboth: the path-length pass adds to the same gradients)
        if g_reg:
            self._g_path_length(gen_z, gen_c, gain, final=sync, noise_unit=pl_noise_unit)
        if not (d_main or d_reg):
            return
        groups = self._merged_groups(gen_z.shape[0], real_img.shape[0]) if d_main and not d_reg else None
        if groups is not None:
            self._d_main_merged(groups, real_img, real_c, gen_z, gen_c, gain, final=sync)
            return
        loss_fake = self._d_fake(gen_z, gen_c, gain) if d_main else 0
        self._d_real(real_img, real_c, gain, final=sync, with_main=d_main, with_r1=d_reg, loss_fake=loss_fake)
