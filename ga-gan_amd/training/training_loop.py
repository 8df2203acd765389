"""Few-shot adaptation training loop: phases (Gmain / Greg / Dmain / Dreg with lazy
regularisation), Adam, generator EMA, and data-parallel gradient exchange.

Implements the behaviour the reference's `training_loop()` is written to have
(training/training_loop.py:163-538; its loop body is mis-indented there and never
reaches the optimizer step): per iteration, every phase whose interval divides the
iteration index zeroes its gradients, enables exactly the parameter groups named by
`requires_grad_parts`, accumulates gradients over the rounds of `batch_gpu` samples,
sanitises them and steps its optimizer; then G_ema is updated.

Multi-GPU: one process per GPU; each rank draws its own latents and takes its own
slice of the real batch; the gradients of the parameters a phase trains are summed
across ranks (RCCL all-reduce over xGMI via torch.distributed) after the last
accumulation round and divided by the world size.  Only tensors that received a
gradient take part, so partially trainable generators (Affine+, AffineLight+) exchange
a few MB instead of the whole network.
"""
import copy
import time
import warnings

import numpy as np
import torch

from .. import images_plugin
from .. import optim_plugin
from ..torch_utils import misc
from ..torch_utils import training_stats
from ..torch_utils.ops import conv2d_gradfix
from . import loss as loss_module
from . import networks


# ----------------------------------------------------------------------------
# Parameter groups of `--generator-requires-grad-parts` (reference: training_loop.py:33-95).

def check_block_resolution(pname, resolution):
    return ('synthesis' in pname) if resolution is None else (f'synthesis.b{resolution}' in pname)


def _rgb(n):
    return 'torgb.weight' in n or 'torgb.bias' in n


name_filters = {
    'mapping': lambda res: lambda n: 'mapping' in n,
    'tRGB_affine': lambda res: lambda n: check_block_resolution(n, res) and 'torgb.affine' in n,
    'tRGB_conv': lambda res: lambda n: check_block_resolution(n, res) and _rgb(n) and 'affine' not in n and 'offset' not in n,
    'tRGB_offset': lambda res: lambda n: check_block_resolution(n, res) and 'torgb.offset' in n and 'torgb.weights_offset' not in n,
    'tRGB_weights_offset': lambda res: lambda n: check_block_resolution(n, res) and 'torgb.weights_offset' in n,
    'tRGB_affine_weights_offset': lambda res: lambda n: check_block_resolution(n, res) and 'torgb.affine.weights_offset' in n,
    'synt_affine': lambda res: lambda n: check_block_resolution(n, res) and 'conv' in n and 'affine' in n,
    'synt_conv': lambda res: lambda n: check_block_resolution(n, res) and 'conv' in n
    and ('weight' in n or 'noise_strength' in n or 'bias' in n) and 'affine' not in n and 'offset' not in n,
    'synt_const': lambda res: lambda n: check_block_resolution(n, res) and 'const' in n,
    'synt_offset': lambda res: lambda n: check_block_resolution(n, res) and 'conv' in n and 'offset' in n and 'weights_offset' not in n,
    'synt_weights_offset': lambda res: lambda n: check_block_resolution(n, res) and 'conv' in n and 'affine' not in n and 'weights_offset' in n,
    'synt_affine_weights_offset': lambda res: lambda n: check_block_resolution(n, res) and 'conv' in n and 'affine.weights_offset' in n,
}


def set_requires_grad_by_filter(module, param_filter, verbose=False):
    for name, param in module.named_parameters():
        if param_filter(name):
            param.requires_grad_(True)
            if verbose:
                print(f'{name: <60}: {str(list(param.shape)): <30} {param.numel():d}')


def set_requires_grad(module, parts, verbose=False):
    """Freeze everything, then enable the groups in `parts` ('all', '<group>' or '<group>.b<res>')."""
    assert isinstance(parts, list)
    module.requires_grad_(False)
    if 'all' in parts:
        module.requires_grad_(True)
        return
    for name, name_filter in name_filters.items():
        for resolution in [1024, 512, 256, 128, 64, 32, 16, 8, 4, None]:
            target = name if resolution is None else f'{name}.b{resolution}'
            if target in parts:
                set_requires_grad_by_filter(module, name_filter(resolution), verbose=verbose)


# ----------------------------------------------------------------------------
# Option presets (reference: train.py:219-279, 449-466).

CFG_SPECS = {
    'stylegan2': dict(ref_gpus=8, kimg=25000, mb=32, mbstd=4, lrate=0.002, gamma=10, ema=10, ramp=None, map=8, fmaps=1),
    'paper256': dict(ref_gpus=8, kimg=25000, mb=64, mbstd=8, lrate=0.0025, gamma=1, ema=20, ramp=None, map=8, fmaps=0.5),
    'paper512': dict(ref_gpus=8, kimg=25000, mb=64, mbstd=8, lrate=0.0025, gamma=0.5, ema=20, ramp=None, map=8, fmaps=1),
    'paper1024': dict(ref_gpus=8, kimg=25000, mb=32, mbstd=4, lrate=0.002, gamma=2, ema=10, ramp=None, map=8, fmaps=1),
}

AFFINE_PLUS = dict(
    generator_requires_grad_parts=['synt_affine', 'tRGB_affine', 'synt_weights_offset.b64', 'tRGB_weights_offset.b64'],
    use_domain_modulation=True, domain_modulation_parametrization='out_in_additive')
AFFINE_LIGHT_PLUS = dict(
    generator_requires_grad_parts=['synt_affine_weights_offset', 'tRGB_affine_weights_offset', 'synt_weights_offset.b64',
                                   'tRGB_weights_offset.b64'],
    use_domain_modulation=True, domain_modulation_parametrization='out_in_additive,affine_out_in_5_1_additive')
FULL = dict(generator_requires_grad_parts=['all'], use_domain_modulation=False, domain_modulation_parametrization=None)


def setup_training_options(cfg='paper256', resolution=256, batch_size=None, num_gpus=1, batch_gpu=None,
                           generator_requires_grad_parts=('all',), use_domain_modulation=False,
                           domain_modulation_parametrization=None, glrate=None, dlrate=None, gamma=None, fp32=False,
                           mbstd=None, c_dim=0, freezed=None):
    """Keyword arguments for `Trainer` the way the reference's `train.py` derives them from its flags."""
    spec = dict(CFG_SPECS[cfg])
    batch_size = spec['mb'] if batch_size is None else batch_size
    batch_gpu = batch_size // num_gpus if batch_gpu is None else batch_gpu
    parts = list(generator_requires_grad_parts)
    synthesis_kwargs = dict(channel_base=int(spec['fmaps'] * 32768), channel_max=512,
                            num_fp16_res=0 if fp32 else 4, conv_clamp=None if fp32 else 256,
                            generator_requires_grad_parts=parts)
    if use_domain_modulation:
        synthesis_kwargs.update(use_domain_modulation=True, domain_modulation_parametrization=domain_modulation_parametrization)
    G_kwargs = dict(z_dim=512, w_dim=512, c_dim=c_dim, img_resolution=resolution, img_channels=3,
                    mapping_kwargs=dict(num_layers=spec['map']), synthesis_kwargs=synthesis_kwargs)
    D_kwargs = dict(c_dim=c_dim, img_resolution=resolution, img_channels=3, channel_base=int(spec['fmaps'] * 32768),
                    channel_max=512, num_fp16_res=0 if fp32 else 4, conv_clamp=None if fp32 else 256,
                    block_kwargs=dict(freeze_layers=freezed) if freezed else {}, mapping_kwargs={},
                    epilogue_kwargs=dict(mbstd_group_size=min(spec['mbstd'] if mbstd is None else mbstd, batch_gpu)))
    lr_g = spec['lrate'] if glrate is None else glrate
    lr_d = spec['lrate'] if dlrate is None else dlrate
    return dict(
        G_kwargs=G_kwargs, D_kwargs=D_kwargs,
        G_opt_kwargs=dict(lr=lr_g, betas=[0, 0.99], eps=1e-8), D_opt_kwargs=dict(lr=lr_d, betas=[0, 0.99], eps=1e-8),
        loss_kwargs=dict(r1_gamma=spec['gamma'] if gamma is None else gamma),
        batch_size=batch_size, batch_gpu=batch_gpu, num_gpus=num_gpus, ema_kimg=spec['ema'], ema_rampup=spec['ramp'],
        requires_grad_parts={'Gboth': parts, 'Gmain': parts, 'Greg': parts, 'Dboth': ['all'], 'Dmain': ['all'], 'Dreg': ['all']})


# ----------------------------------------------------------------------------
# Gradient exchange.

class GradReducer:
    """Average the gradients of a phase's parameters across ranks with as few collectives as possible and no copies:
    the gradients LIVE in flat buffers (one per bucket of `bucket_bytes`), so a bucket is all-reduced where it lies.

        reducer.attach(all parameters)            once: completion hooks (idle until armed)
        reducer.prepare(trainable parameters)     per phase, instead of zero_grad(): every `.grad` becomes a zeroed view
                                                  into its bucket; autograd accumulates into it in place
        ... backward calls ...                    `arm()` right before the one whose gradients are final (the call the
                                                  reference synchronises on: `sync` in training/loss.py:69-152, misc.ddp_sync)
        reducer.finish(parameters)                wait, average; parameters no backward call reached get `.grad = None`

    While armed, every parameter reports from autograd's post-accumulate hook the moment its gradient is complete; when
    the last member of a bucket has reported, the bucket's all-reduce is enqueued (asynchronously, on the backend's
    own stream), so the exchange of the layers that finish first runs under the backward pass of the remaining
    ones.  Buckets are cut in reverse parameter order - the order in which a backward pass completes them.  The
    discriminator profits most: its 512-channel low-resolution weights, most of its bytes, are the first to finish.
    All ranks run the same graph, so every rank enqueues the same buckets in the same order.

    Gradients that are not views of a bucket (a caller that never called `prepare`, e.g. a phase replayed from a
    HIP graph, whose gradients live in the graph's pool) are still exchanged by `finish` / `reducer(params)`: packed into
    a temporary flat buffer per bucket and copied back.
    """

    def __init__(self, world_size, bucket_bytes=64 << 20, group=None):
        self.world_size = world_size
        self.bucket_bytes = bucket_bytes
        self.group = group
        self._attached = set()
        self._armed = False
        self._plans = {}                 # tuple(id(p)) -> plan
        self._plan = None                # the plan of the phase in progress
        self._touched, self._reported = set(), set()
        self._pending = []               # [(work, flat buffer, None | [loose gradient tensors])]
        self.collectives = 0             # all-reduce calls issued so far (tests, diagnostics)

    # -- the plan of a parameter list ------------------------------------------------
    class _Plan:
        def __init__(self, params, bucket_bytes):
            self.params = list(params)
            self.buffers, self.bucket_of, self.members, self.views = [], {}, [], {}
            order = list(reversed(self.params))            # completion order of a backward pass
            start = 0
            while start < len(order):
                end, size = start, 0
                while end < len(order) and order[end].dtype == order[start].dtype and \
                        (end == start or size + order[end].numel() * order[end].element_size() <= bucket_bytes):
                    size += order[end].numel() * order[end].element_size()
                    end += 1
                group = order[start:end]
                flat = torch.zeros(sum(q.numel() for q in group), dtype=group[0].dtype, device=group[0].device)
                offset = 0
                for q in group:
                    self.views[id(q)] = flat[offset:offset + q.numel()].view(q.shape)
                    self.bucket_of[id(q)] = len(self.buffers)
                    offset += q.numel()
                self.buffers.append(flat)
                self.members.append({id(q) for q in group})
                start = end

    def prepare(self, params):
        """Instead of `zero_grad()` at the start of a phase: point every parameter's `.grad` into its bucket, zeroed."""
        params = [p for p in params]
        if self.world_size <= 1:
            for p in params:
                p.grad = None
            return
        key = tuple(id(p) for p in params)
        plan = self._plans.get(key)
        if plan is None or any(plan.views[id(p)].device != p.device for p in params):
            plan = self._plans[key] = self._Plan(params, self.bucket_bytes)
        for flat in plan.buffers:
            flat.zero_()
        for p in params:
            p.grad = plan.views[id(p)]
        self._plan = plan
        self._launched = [False] * len(plan.buffers)
        self._waiting = [set(m) for m in plan.members]
        self._touched, self._reported = set(), set()

    # -- hooks -------------------------------------------------------------------------
    def attach(self, params):
        """Register the completion hook on every parameter (once; idle until `arm()`)."""
        if self.world_size <= 1:
            return
        for p in params:
            if id(p) not in self._attached:
                self._attached.add(id(p))
                frozen = not p.requires_grad          # (hooks can only be registered on tensors that require grad;
                p.requires_grad_(True)                #  the hook stays when the flag is switched back)
                p.register_post_accumulate_grad_hook(self._on_grad_ready)
                p.requires_grad_(not frozen)

    def arm(self):
        """The next backward call produces final gradients: exchange them as they complete."""
        if self.world_size > 1:
            self._armed = True
            self._reported = set()

    def _on_grad_ready(self, p):
        self._touched.add(id(p))
        if not self._armed or p.grad is None:
            return
        self._reported.add(id(p))
        plan = self._plan
        if plan is None or id(p) not in plan.views or p.grad.data_ptr() != plan.views[id(p)].data_ptr():
            return                                    # a loose gradient: finish() packs it
        b = plan.bucket_of[id(p)]
        self._waiting[b].discard(id(p))
        if not self._waiting[b] and not self._launched[b]:
            self._launch(b)

    def _all_reduce(self, flat):
        self.collectives += 1
        return torch.distributed.all_reduce(flat, group=self.group, async_op=True)

    def _launch(self, b):
        self._launched[b] = True
        self._pending.append((self._all_reduce(self._plan.buffers[b]), self._plan.buffers[b], None))

    # -- end of the phase -----------------------------------------------------------------
    def finish(self, params):
        """After the phase's last backward call: enqueue the buckets that are not under way yet, exchange gradients
        that do not live in a bucket, wait, average.  Returns the bytes exchanged."""
        if self.world_size <= 1:
            return 0
        self._armed = False
        plan, loose = self._plan, []
        if plan is not None:
            for b in range(len(plan.buffers)):
                if not self._launched[b]:
                    self._launch(b)
        for p in params:
            if p.grad is None:
                continue
            in_bucket = plan is not None and id(p) in plan.views and p.grad.data_ptr() == plan.views[id(p)].data_ptr()
            if not in_bucket:
                loose.append(p.grad)
        self._exchange_loose(loose)
        total = self._drain()
        if plan is not None:
            for p in plan.params:          # a parameter that no backward call of the phase reached has no gradient
                if id(p) not in self._touched and p.grad is not None and p.grad.data_ptr() == plan.views[id(p)].data_ptr():
                    p.grad = None
        self._plan = None
        return total

    def __call__(self, params):
        """Exchange the gradients of `params` as they are (after the backward pass, no hooks, no prepared buckets)."""
        if self.world_size <= 1:
            return 0
        self._exchange_loose([p.grad for p in params if p.grad is not None])
        return self._drain()

    def _exchange_loose(self, grads):
        start = 0
        while start < len(grads):
            end, size = start, 0
            while end < len(grads) and grads[end].dtype == grads[start].dtype and \
                    (end == start or size + grads[end].numel() * grads[end].element_size() <= self.bucket_bytes):
                size += grads[end].numel() * grads[end].element_size()
                end += 1
            bucket = grads[start:end]
            flat = torch.cat([g.reshape(-1) for g in bucket])
            self._pending.append((self._all_reduce(flat), flat, bucket))
            start = end

    def _drain(self):
        pending, self._pending = self._pending, []
        total = 0
        for work, flat, bucket in pending:
            work.wait()
            flat.div_(self.world_size)
            if bucket is not None:
                offset = 0
                for g in bucket:
                    g.copy_(flat[offset:offset + g.numel()].view_as(g))
                    offset += g.numel()
            total += flat.numel() * flat.element_size()
        return total


# ----------------------------------------------------------------------------

class Phase:
    def __init__(self, name, module, opt, interval):
        self.name, self.module, self.opt, self.interval = name, module, opt, interval


# Augmentation presets of the reference's `--augpipe` flag (train.py:357-382).
_BLIT = dict(xflip=1, rotate90=1, xint=1)
_GEOM = dict(scale=1, rotate=1, aniso=1, xfrac=1)
_COLOR = dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)
AUGPIPE_SPECS = {
    'blit': dict(_BLIT), 'geom': dict(_GEOM), 'color': dict(_COLOR), 'filter': dict(imgfilter=1), 'noise': dict(noise=1),
    'cutout': dict(cutout=1), 'bg': dict(**_BLIT, **_GEOM), 'bgc': dict(**_BLIT, **_GEOM, **_COLOR),
    'bgcf': dict(**_BLIT, **_GEOM, **_COLOR, imgfilter=1), 'bgcfn': dict(**_BLIT, **_GEOM, **_COLOR, imgfilter=1, noise=1),
    'bgcfnc': dict(**_BLIT, **_GEOM, **_COLOR, imgfilter=1, noise=1, cutout=1),
}


def make_augment_pipe(augpipe='bgc', augment_p=0.0, device=None):
    """The frozen AugmentPipe of a preset with its strength set (reference: training_loop.py:259-266)."""
    from . import augment
    pipe = augment.AugmentPipe(**AUGPIPE_SPECS[augpipe]).train().requires_grad_(False)
    if device is not None:
        pipe = pipe.to(device)
    pipe.p.copy_(torch.as_tensor(float(augment_p)))
    return pipe


class Trainer:
    """State and one-iteration step of the training loop, for one rank.

    `augment_pipe` (an AugmentPipe) is applied in front of every discriminator call; with `ada_target` its strength
    follows the reference's heuristic (training_loop.py:529-535): every `ada_interval` iterations the mean sign of
    the discriminator's logits on real images is compared with the target and `p` moves by
    batch_size * ada_interval / (ada_kimg * 1000) towards it (never below 0)."""

    def __init__(self, G_kwargs, D_kwargs, G_opt_kwargs, D_opt_kwargs, loss_kwargs=None, batch_size=32, batch_gpu=32,
                 num_gpus=1, rank=0, ema_kimg=10, ema_rampup=None, G_reg_interval=4, D_reg_interval=16,
                 requires_grad_parts=None, device=None, random_seed=0, augment_pipe=None, resume=None, use_graphs=False,
                 ada_target=None, ada_interval=4, ada_kimg=500):
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.rank, self.num_gpus = rank, num_gpus
        self.batch_size, self.batch_gpu = batch_size, batch_gpu
        assert batch_size % (batch_gpu * num_gpus) == 0
        self.ema_kimg, self.ema_rampup = ema_kimg, ema_rampup
        np.random.seed(random_seed * num_gpus + rank)
        torch.manual_seed(random_seed * num_gpus + rank)
        conv2d_gradfix.enabled = True

        # All ranks build identical networks (the constructors draw from a fixed seed).
        rng_state = torch.get_rng_state()
        torch.manual_seed(random_seed)
        self.G = networks.Generator(**G_kwargs).train().requires_grad_(False).to(self.device)
        self.D = networks.Discriminator(**D_kwargs).train().requires_grad_(False).to(self.device)
        torch.set_rng_state(rng_state)
        if resume is not None:
            for name, module in (('G', self.G), ('D', self.D)):
                if name in resume:
                    module.load_state_dict(resume[name], strict=False)
        self.G_ema = copy.deepcopy(self.G).eval()
        if resume is not None and 'G_ema' in resume:
            self.G_ema.load_state_dict(resume['G_ema'], strict=False)

        self.loss = loss_module.StyleGAN2Loss(device=self.device, G_mapping=self.G.mapping,
                                              G_synthesis=self.G.synthesis, D=self.D, augment_pipe=augment_pipe,
                                              **(loss_kwargs or {}))
        self.requires_grad_parts = requires_grad_parts or {k: ['all'] for k in
                                                           ('Gboth', 'Gmain', 'Greg', 'Dboth', 'Dmain', 'Dreg')}
        self.augment_pipe = augment_pipe
        self.ada_target, self.ada_interval, self.ada_kimg = ada_target, ada_interval, ada_kimg
        self.ada_stats = None
        if augment_pipe is not None:
            from ..torch_utils.ops import grid_sample_gradfix
            grid_sample_gradfix.enabled = True     # R1 differentiates twice through the resampling (training_loop.py:210)
            if ada_target is not None:
                training_stats.enable(True)
                self.ada_stats = training_stats.Collector(regex='Loss/signs/real')
        self.reducer = GradReducer(num_gpus)
        self.reducer.attach(list(self.G.parameters()) + list(self.D.parameters()))
        # HIP graphs: the forward/backward of a phase is recorded once (after its first eager run) and replayed
        # afterwards, which removes the launch cost of its few thousand small kernels from the host.  Every phase
        # runs the augmentation pipe in front of the discriminator, and the pipe draws its per-image parameters and
        # padding sizes on the host: a recording would freeze them (and the strength p), so with a pipe the phases
        # are launched from the host as usual.
        if use_graphs and augment_pipe is not None:
            warnings.warn('use_graphs is ignored while an augment_pipe is set: the pipe draws its parameters on the host')
            use_graphs = False
        self.use_graphs = bool(use_graphs)
        self.loss.grad_sync = None if self.use_graphs else self.reducer   # (a recorded phase exchanges after its replay)
        self.phases = []
        for name, module, opt_kwargs, reg_interval in (('G', self.G, G_opt_kwargs, G_reg_interval),
                                                       ('D', self.D, D_opt_kwargs, D_reg_interval)):
            opt_kwargs = dict(opt_kwargs)
            if reg_interval is None:
                opt = optim_plugin.FusedAdam(module.parameters(), **opt_kwargs)
                self.phases.append(Phase(name + 'both', module, opt, 1))
            else:   # lazy regularisation: compensate the optimizer for the extra steps
                mb_ratio = reg_interval / (reg_interval + 1)
                opt_kwargs['lr'] = opt_kwargs['lr'] * mb_ratio
                opt_kwargs['betas'] = [beta ** mb_ratio for beta in opt_kwargs['betas']]
                opt = optim_plugin.FusedAdam(module.parameters(), **opt_kwargs)
                self.phases.append(Phase(name + 'main', module, opt, 1))
                self.phases.append(Phase(name + 'reg', module, opt, reg_interval))
        self.cur_nimg = 0
        self.batch_idx = 0
        self.exchanged_bytes = 0
        self._graphs = {}

    # -- snapshots ---------------------------------------------------------------
    def snapshot(self):
        """Everything a run needs to continue, as plain tensors and numbers: the three networks (the reference's
        snapshot, training_loop.py:586-601, holds these) and what the reference loses on resume - optimizer
        moments and step counts, the path-length average, the augmentation strength, the image counter."""
        opts = {}
        for phase in self.phases:
            opts.setdefault(phase.name[0], phase.opt.state_dict())       # main / reg phases share one optimizer
        state = dict(G=self.G.state_dict(), D=self.D.state_dict(), G_ema=self.G_ema.state_dict(), opt=opts,
                     pl_mean=self.loss.pl_mean.detach().clone(), cur_nimg=self.cur_nimg, batch_idx=self.batch_idx)
        if self.augment_pipe is not None:
            state['augment_pipe'] = self.augment_pipe.state_dict()
        return state

    def load_snapshot(self, state):
        self.G.load_state_dict(state['G'])
        self.D.load_state_dict(state['D'])
        self.G_ema.load_state_dict(state['G_ema'])
        done = set()
        for phase in self.phases:
            if phase.name[0] in state.get('opt', {}) and phase.name[0] not in done:
                phase.opt.load_state_dict(state['opt'][phase.name[0]])
                done.add(phase.name[0])
        self.loss.pl_mean.copy_(state['pl_mean'].to(self.device))
        self.cur_nimg, self.batch_idx = int(state['cur_nimg']), int(state['batch_idx'])
        if self.augment_pipe is not None and 'augment_pipe' in state:
            self.augment_pipe.load_state_dict(state['augment_pipe'])

    def save(self, path):
        torch.save(self.snapshot(), path)

    def save_network_pkl(self, path, training_set_kwargs=None):
        """The reference's `network-snapshot-*.pkl` (training_loop.py:586-601): G, D, G_ema and the augmentation pipe as
        evaluation-mode copies without gradients on the CPU, in the record layout its `legacy.load_network_pkl` and
        this package's read."""
        from .. import legacy
        data = dict(training_set_kwargs=dict(training_set_kwargs or {}))
        for name, module in (('G', self.G), ('D', self.D), ('G_ema', self.G_ema), ('augment_pipe', self.augment_pipe)):
            if module is not None:
                module = copy.deepcopy(module).eval().requires_grad_(False).cpu()
            data[name] = module
        with open(path, 'wb') as f:
            legacy.save_network_pkl(f, data)

    def load(self, path):
        """Tensors, numbers and containers only: read with `weights_only=True` (nothing in the file is executed)."""
        self.load_snapshot(torch.load(path, map_location=self.device, weights_only=True))

    # -- data ------------------------------------------------------------------
    def synthetic_batch(self, generator=None):
        """Uniform random uint8 images of this rank's share of the batch (and empty labels)."""
        n = self.batch_size // self.num_gpus
        res = self.G.img_resolution
        img = torch.randint(0, 256, [n, self.G.img_channels, res, res], dtype=torch.uint8, device=self.device,
                            generator=generator)
        return img, torch.zeros([n, self.G.c_dim], device=self.device)

    # -- HIP graphs ---------------------------------------------------------------
    def _capture_phase(self, phase, r_img, r_c, g_z, g_c):
        """Record the forward/backward of `phase` on static copies of its inputs.  Nothing executes during the
        recording; the gradients it produces live in the graph's memory pool and are handed to the parameters
        after every replay."""
        static = dict(r_img=r_img.clone(), r_c=r_c.clone(), g_z=g_z.clone(), g_c=g_c.clone())
        phase.opt.zero_grad(set_to_none=True)
        set_requires_grad(phase.module, parts=self.requires_grad_parts[phase.name])
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):   # a memory pool of its own: the gradients it holds outlive every replay
            self.loss.accumulate_gradients(phase=phase.name, real_img=static['r_img'], real_c=static['r_c'],
                                           gen_z=static['g_z'], gen_c=static['g_c'], sync=True, gain=phase.interval)
        grads = [(p, p.grad) for p in phase.module.parameters() if p.grad is not None]
        phase.module.requires_grad_(False)
        phase.opt.zero_grad(set_to_none=True)
        self._graphs[phase.name] = (graph, static, grads)

    def _replay_phase(self, phase, r_img, r_c, g_z, g_c):
        graph, static, grads = self._graphs[phase.name]
        static['r_img'].copy_(r_img)
        static['r_c'].copy_(r_c)
        static['g_z'].copy_(g_z)
        static['g_c'].copy_(g_c)
        phase.opt.zero_grad(set_to_none=True)
        graph.replay()
        for p, g in grads:
            p.grad = g

    # -- one iteration ----------------------------------------------------------
    def step(self, real_img, real_c, gen_z=None, gen_c=None):
        """real_img: this rank's share, uint8 [n, C, H, W] (or float already in [-1, 1]).  gen_z / gen_c: latents and
        labels of the generated images, [phases * n, ...] (one run of n per phase).  The reference draws every
        generated label from a random item of the dataset (training_loop.py:364-366); a caller with a label table
        passes such draws as gen_c.  Without gen_c the labels of this real batch are resampled with replacement,
        independently for every phase."""
        n_rank = self.batch_size // self.num_gpus
        with torch.autograd.profiler.record_function('data_fetch'):
            real = real_img.to(self.device)
            if real.dtype == torch.uint8 and real.ndim == 4:
                real = images_plugin.images_to_float(real)
            elif not real.is_floating_point():
                real = real.to(torch.float32) / 127.5 - 1
            phase_real_img = real.split(self.batch_gpu)
            phase_real_c = real_c.to(self.device).split(self.batch_gpu)
            if gen_z is None:
                gen_z = torch.randn([len(self.phases) * n_rank, self.G.z_dim], device=self.device)
            all_gen_z = [z.split(self.batch_gpu) for z in gen_z.split(n_rank)]
            if gen_c is None:
                gen_c = real_c.to(self.device)
                if gen_c.shape[1] > 0:
                    gen_c = gen_c[torch.randint(gen_c.shape[0], [len(self.phases) * n_rank], device=self.device)]
                else:
                    gen_c = gen_c.new_zeros([len(self.phases) * n_rank, 0])
            all_gen_c = [c.split(self.batch_gpu) for c in gen_c.to(self.device).split(n_rank)]
        rounds = n_rank // self.batch_gpu

        for phase, phase_gen_z, phase_gen_c in zip(self.phases, all_gen_z, all_gen_c):
            if self.batch_idx % phase.interval != 0:
                continue
            graphed = self.use_graphs and rounds == 1 and self._graphs.get(phase.name)
            if graphed:
                self._replay_phase(phase, phase_real_img[0], phase_real_c[0], phase_gen_z[0], phase_gen_c[0])
            else:
                phase.opt.zero_grad(set_to_none=True)
                set_requires_grad(phase.module, parts=self.requires_grad_parts[phase.name])
                if self.num_gpus > 1:     # gradients accumulate straight into the buckets that are exchanged
                    self.reducer.prepare([p for p in phase.module.parameters() if p.requires_grad])
                for round_idx, (r_img, r_c, g_z, g_c) in enumerate(zip(phase_real_img, phase_real_c, phase_gen_z, phase_gen_c)):
                    self.loss.accumulate_gradients(phase=phase.name, real_img=r_img, real_c=r_c, gen_z=g_z, gen_c=g_c,
                                                   sync=(round_idx == rounds - 1), gain=phase.interval)
                phase.module.requires_grad_(False)
            with torch.autograd.profiler.record_function(phase.name + '_opt'):
                params = [p for p in phase.module.parameters() if p.grad is not None]
                self.exchanged_bytes += self.reducer.finish(params)
                if params:
                    phase.opt.step()   # (sanitises the gradients: NaN -> 0, +-Inf -> +-1e5, then Adam; one launch)
            if self.use_graphs and rounds == 1 and phase.name not in self._graphs and params:   # (a phase that does nothing is not recorded)
                self._capture_phase(phase, phase_real_img[0], phase_real_c[0], phase_gen_z[0], phase_gen_c[0])

        with torch.autograd.profiler.record_function('Gema'):
            ema_nimg = self.ema_kimg * 1000
            if self.ema_rampup is not None:
                ema_nimg = min(ema_nimg, self.cur_nimg * self.ema_rampup)
            ema_beta = 0.5 ** (self.batch_size / max(ema_nimg, 1e-8))
            with torch.no_grad():
                ema_params, params = list(self.G_ema.parameters()), list(self.G.parameters())
                optim_plugin.ema_(ema_params, params, ema_beta)            # p_ema = p.lerp(p_ema, ema_beta)
                torch._foreach_copy_(list(self.G_ema.buffers()), list(self.G.buffers()))
        self.cur_nimg += self.batch_size
        self.batch_idx += 1
        if self.ada_stats is not None and self.batch_idx % self.ada_interval == 0:
            self.ada_stats.update()     # (all-reduces the sign statistics across ranks)
            sign_real = self.ada_stats['Loss/signs/real']
            if sign_real == sign_real:   # not NaN: at least one real batch was seen since the last update
                adjust = np.sign(sign_real - self.ada_target) * (self.batch_size * self.ada_interval) / (self.ada_kimg * 1000)
                self.augment_pipe.p.copy_((self.augment_pipe.p + adjust).clamp_min(0))


def training_loop(total_kimg=1, kimg_per_tick=4, data_iterator=None, progress_fn=None, rank=0, verbose=True,
                  **trainer_kwargs):
    """Run `Trainer.step` until `total_kimg` thousand real images have been shown.  `data_iterator`
    yields (uint8 images, labels) for this rank; None = synthetic images.  Prints one status line per
    tick (`Timing/sec_per_kimg` as the reference's log does) and returns the trainer."""
    trainer = Trainer(rank=rank, **trainer_kwargs)
    tick_start_nimg, tick_start_time, tick = 0, time.time(), 0
    start_time = tick_start_time
    while trainer.cur_nimg < total_kimg * 1000:
        real_img, real_c = next(data_iterator) if data_iterator is not None else trainer.synthetic_batch()
        trainer.step(real_img, real_c)
        done = trainer.cur_nimg >= total_kimg * 1000
        if done or trainer.cur_nimg >= tick_start_nimg + kimg_per_tick * 1000:
            torch.cuda.synchronize(trainer.device)
            now = time.time()
            sec_per_kimg = (now - tick_start_time) / max(trainer.cur_nimg - tick_start_nimg, 1) * 1e3
            training_stats.report('Timing/sec_per_kimg', sec_per_kimg)
            if rank == 0 and verbose:
                print(f'tick {tick:<5d} kimg {trainer.cur_nimg / 1e3:<8.1f} time {now - start_time:<8.1f} '
                      f'sec/kimg {sec_per_kimg:<8.2f}', flush=True)
            tick += 1
            tick_start_nimg, tick_start_time = trainer.cur_nimg, now
            if progress_fn is not None:
                progress_fn(trainer.cur_nimg // 1000, total_kimg)
    return trainer
