#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference; CPU only).  It imports
the reference's own Python modules from /root/reference/DissimilarDomains,
evaluates them on small seeded inputs and stores inputs + outputs as .npz
files.  The fixtures are data (arrays plus a JSON description of each case);
nothing of the reference's source travels.

The reference's `SynthesisBlock.forward` crashes on the first block (it
up-samples the image skip before one exists); the harness swaps in the
upstream block wiring below, without touching /root/reference.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import io
import json
import math
import os
import sys
import zlib
import contextlib

import numpy as np
import torch

REF = '/root/reference/DissimilarDomains'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from torch_utils.ops import upfirdn2d as ref_upfirdn2d  # noqa: E402
from torch_utils.ops import bias_act as ref_bias_act  # noqa: E402
from torch_utils.ops import conv2d_resample as ref_conv2d_resample  # noqa: E402
from torch_utils.ops import fma as ref_fma  # noqa: E402
from torch_utils import misc as ref_misc  # noqa: E402
from training import networks as ref_networks  # noqa: E402
from training import loss as ref_loss  # noqa: E402
from training import training_loop as ref_training_loop  # noqa: E402


# ----------------------------------------------------------------------------
# Upstream block wiring for SynthesisBlock (image skip up-sampled only if present).

def _block_forward(self, x, img, ws, force_fp32=False, fused_modconv=None, **layer_kwargs):
    w_iter = iter(ws.unbind(dim=1))
    dtype = torch.float16 if self.use_fp16 and not force_fp32 else torch.float32
    if fused_modconv is None:
        fused_modconv = (not self.training) and (dtype == torch.float32 or int(ws.shape[0]) == 1)
    if self.in_channels == 0:
        x = self.const.to(dtype=dtype).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
        x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
    else:
        x = x.to(dtype=dtype)
        if self.architecture == 'resnet':
            y = self.skip(x, gain=np.sqrt(0.5))
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, gain=np.sqrt(0.5), **layer_kwargs)
            x = y.add_(x)
        else:
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
    if img is not None:
        img = ref_upfirdn2d.upsample2d(img, self.resample_filter)
    if self.is_last or self.architecture == 'skip':
        y = self.torgb(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        y = y.to(dtype=torch.float32)
        img = img.add_(y) if img is not None else y
    return x, img


ref_networks.SynthesisBlock.forward = _block_forward


# ----------------------------------------------------------------------------

def npy(t):
    return t.detach().cpu().numpy().copy()


class Bundle:
    """Collects cases: arrays under 'c<i>_<name>' plus one JSON list of metadata."""

    def __init__(self, name):
        self.name = name
        self.arrays = {}
        self.meta = []

    def add(self, meta, **arrays):
        i = len(self.meta)
        self.meta.append(meta)
        for k, v in arrays.items():
            if v is None:
                continue
            self.arrays[f'c{i}_{k}'] = npy(v) if isinstance(v, torch.Tensor) else np.asarray(v)

    def save(self):
        path = os.path.join(OUT, self.name + '.npz')
        np.savez_compressed(path, meta=np.frombuffer(json.dumps(self.meta).encode(), dtype=np.uint8), **self.arrays)
        print(f'{path}: {len(self.meta)} cases, {os.path.getsize(path) / 1024:.0f} KiB')


def rnd(gen, *shape, dtype=torch.float32):
    return torch.randn(*shape, generator=gen, dtype=dtype)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


# ----------------------------------------------------------------------------
# Ops.

def gen_upfirdn2d():
    g = torch.Generator().manual_seed(100)
    b = Bundle('upfirdn2d')
    f44 = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    f12 = ref_upfirdn2d.setup_filter(  # 12-tap separable (sym6 wavelet low-pass, as used by the ADA pipe)
        [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
         0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
         0.0017677118642428036, -0.007800708325034148])
    f3 = ref_upfirdn2d.setup_filter([1, 2, 1])
    f35 = rnd(g, 3, 5)
    filters = {'f44': f44, 'f12': f12, 'none': None, 'f3': f3, 'f35': f35}
    cases = []
    for shape in ([2, 3, 16, 16], [2, 8, 9, 11]):
        for fname in ('f44', 'none', 'f12', 'f35'):
            for up in (1, 2):
                for down in (1, 2):
                    for pad in (0, [2, 1, 2, 1], [-1, 2, 3, -2], [5, 4, 6, 7]):
                        for flip in (False, True):
                            cases.append((shape, fname, up, down, pad, flip, 1.0 if not flip else 2.5))
    cases += [([1, 2, 7, 5], 'f3', [2, 1], [1, 3], [1, 0, 2, 2], False, 1.0),
              ([1, 2, 7, 5], 'f44', [1, 3], [2, 1], [3, 3, 1, 4], True, 0.5),
              ([1, 1, 1, 1], 'f44', 2, 1, [2, 1, 2, 1], False, 4.0),
              ([2, 5, 33, 31], 'f44', 1, 1, [1, 1, 1, 1], False, 4.0),     # blur after transposed conv
              ([2, 3, 16, 16], 'f44', 2, 1, [2, 1, 2, 1], False, 4.0),     # image-skip up-sampling
              ([2, 4, 32, 32], 'f44', 1, 1, [2, 2, 2, 2], False, 1.0),     # pre-blur of the strided conv
              ([2, 4, 32, 32], 'f44', 1, 2, [1, 1, 1, 1], False, 1.0)]     # skip-branch decimation
    kept = 0
    for shape, fname, up, down, pad, flip, gain in cases:
        f = filters[fname]
        x = rnd(g, *shape).requires_grad_(True)
        try:
            y = ref_upfirdn2d.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain, impl='ref')
        except Exception:
            continue  # empty output for this combination
        if y.numel() == 0:
            continue
        dy = rnd(g, *y.shape).requires_grad_(True)
        gx, = torch.autograd.grad((y * dy).sum(), x, create_graph=True)
        v = rnd(g, *gx.shape)
        ggdy, = torch.autograd.grad((gx * v).sum(), dy)
        # thin the cross product: keep every third generic case, and all named ones
        kept += 1
        if len(b.meta) >= 0 and (kept % 3 != 0) and shape in ([2, 3, 16, 16], [2, 8, 9, 11]) and gain in (1.0, 2.5):
            continue
        b.add(dict(filter=fname, up=up, down=down, padding=pad, flip_filter=flip, gain=gain),
              x=x, f=f, y=y, dy=dy, gx=gx, v=v, ggdy=ggdy)
    b.save()


def gen_bias_act():
    g = torch.Generator().manual_seed(101)
    b = Bundle('bias_act')
    s2 = math.sqrt(2)
    cases = []
    for act in ('linear', 'lrelu'):
        for gain in (None, 1.0, s2, s2 * math.sqrt(0.5)):
            for clamp in (None, 256.0, 0.5):
                cases.append((act, [2, 5, 6, 7], 1, True, None, gain, clamp))
                cases.append((act, [4, 9], 1, True, None, gain, clamp))
    for act in ('relu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish'):
        cases.append((act, [2, 4, 5, 3], 1, True, None, None, None))
        cases.append((act, [2, 4, 5, 3], 1, True, None, 1.7, 0.8))
    cases += [('lrelu', [3, 4, 5], 2, True, 0.1, 2.0, None), ('lrelu', [3, 4, 5], 0, True, None, None, 1.0),
              ('lrelu', [2, 3, 4, 4], 1, False, None, None, None), ('linear', [2, 3, 4, 4], 3, True, None, 3.0, 2.0)]
    for act, shape, dim, has_b, alpha, gain, clamp in cases:
        x = (rnd(g, *shape) * 1.5).requires_grad_(True)
        bias = rnd(g, shape[dim]).requires_grad_(True) if has_b else None
        y = ref_bias_act.bias_act(x, bias, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp, impl='ref')
        dy = rnd(g, *shape).requires_grad_(True)
        ins = [x] + ([bias] if has_b else [])
        grads = torch.autograd.grad((y * dy).sum(), ins, create_graph=True)
        gx = grads[0]
        gb = grads[1] if has_b else None
        v = rnd(g, *shape)
        second = torch.autograd.grad((gx * v).sum(), [dy, x], allow_unused=True)
        b.add(dict(act=act, dim=dim, alpha=alpha, gain=gain, clamp=clamp),
              x=x, b=bias, y=y, dy=dy, gx=gx, gb=gb, v=v, ggdy=second[0],
              ggx=second[1] if second[1] is not None else torch.zeros_like(x))
    b.save()


def gen_conv2d_resample():
    g = torch.Generator().manual_seed(102)
    b = Bundle('conv2d_resample')
    f = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    cases = [  # (x shape, w shape, up, down, padding, groups, flip_weight, with filter)
        ([2, 6, 16, 16], [4, 6, 1, 1], 1, 2, 0, 1, True, True),     # 1x1 + down
        ([2, 6, 8, 8], [4, 6, 1, 1], 2, 1, 0, 1, True, True),       # 1x1 + up
        ([2, 6, 16, 16], [4, 6, 3, 3], 1, 2, 1, 1, True, True),     # blur + strided conv
        ([2, 6, 8, 8], [4, 6, 3, 3], 2, 1, 1, 1, False, True),      # transposed conv + blur
        ([2, 6, 8, 8], [4, 6, 3, 3], 2, 1, 1, 1, True, True),
        ([1, 6, 8, 8], [8, 3, 3, 3], 2, 1, 1, 2, False, True),      # grouped up-sampling
        ([2, 6, 12, 10], [4, 6, 3, 3], 1, 1, 1, 1, True, False),    # plain conv
        ([1, 6, 12, 10], [9, 2, 3, 3], 1, 1, 1, 3, True, False),    # grouped plain conv
        ([2, 6, 12, 10], [4, 6, 3, 3], 1, 1, [1, 2, 0, 1], 1, True, False),   # generic fallback
        ([2, 6, 8, 8], [4, 6, 3, 3], 2, 2, 1, 1, True, True),       # up and down
        ([2, 6, 9, 9], [4, 6, 3, 3], 1, 1, 1, 1, False, False),     # true convolution
    ]
    for xs, ws, up, down, pad, groups, flip_weight, with_f in cases:
        x = rnd(g, *xs).requires_grad_(True)
        w = (rnd(g, *ws) * 0.3).requires_grad_(True)
        y = ref_conv2d_resample.conv2d_resample(x, w, f=f if with_f else None, up=up, down=down, padding=pad,
                                                groups=groups, flip_weight=flip_weight)
        dy = rnd(g, *y.shape)
        gx, gw = torch.autograd.grad((y * dy).sum(), [x, w])
        b.add(dict(up=up, down=down, padding=pad, groups=groups, flip_weight=flip_weight, with_filter=with_f),
              x=x, w=w, f=f if with_f else None, y=y, dy=dy, gx=gx, gw=gw)
    b.save()


def gen_modulated_conv2d():
    g = torch.Generator().manual_seed(103)
    b = Bundle('modulated_conv2d')
    f = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    for demod in (True, False):
        for up in (1, 2):
            for fused in (True, False):
                for noise_kind in ('none', 'const', 'batch'):
                    for k in (3, 1):
                        if k == 1 and (up == 2 or noise_kind != 'none'):
                            continue
                        x = rnd(g, 2, 8, 16, 16).requires_grad_(True)
                        w = rnd(g, 4, 8, k, k).requires_grad_(True)
                        s = (rnd(g, 2, 8) + 1).requires_grad_(True)
                        res = 16 * up
                        noise = None
                        if noise_kind == 'const':
                            noise = rnd(g, res, res) * 0.3
                        elif noise_kind == 'batch':
                            noise = rnd(g, 2, 1, res, res) * 0.3
                        y = ref_networks.modulated_conv2d(
                            x=x, weight=w, styles=s, noise=noise, up=up, padding=k // 2, resample_filter=f,
                            demodulate=demod, flip_weight=(up == 1), fused_modconv=fused)
                        dy = rnd(g, *y.shape)
                        gx, gw, gs = torch.autograd.grad((y * dy).sum(), [x, w, s])
                        b.add(dict(demodulate=demod, up=up, fused_modconv=fused, noise=noise_kind, kernel=k),
                              x=x, w=w, s=s, noise=noise, y=y, dy=dy, gx=gx, gw=gw, gs=gs)
    b.save()


def gen_misc_ops():
    g = torch.Generator().manual_seed(104)
    b = Bundle('misc_ops')
    for n, grp, c, f in ((4, 4, 16, 1), (8, 4, 16, 1), (8, 8, 12, 2), (4, 8, 8, 1), (6, None, 8, 1)):
        x = rnd(g, n, c, 4, 4).requires_grad_(True)
        y = ref_networks.MinibatchStdLayer(group_size=grp, num_channels=f)(x)
        dy = rnd(g, *y.shape)
        gx, = torch.autograd.grad((y * dy).sum(), x, create_graph=True)
        v = rnd(g, *x.shape)
        ggx, = torch.autograd.grad((gx * v).sum(), x)
        b.add(dict(op='mbstd', group_size=grp, num_channels=f), x=x, y=y, dy=dy, gx=gx, v=v, ggx=ggx)
    a = rnd(g, 2, 4, 6, 6).requires_grad_(True)
    bb = rnd(g, 2, 4, 1, 1).requires_grad_(True)
    c = rnd(g, 2, 1, 6, 6).requires_grad_(True)
    y = ref_fma.fma(a, bb, c)
    dy = rnd(g, *y.shape)
    ga, gb, gc = torch.autograd.grad((y * dy).sum(), [a, bb, c])
    b.add(dict(op='fma'), a=a, b=bb, c=c, y=y, dy=dy, ga=ga, gb=gb, gc=gc)
    b.save()


def gen_grid_sample():
    """grid_sample_gradfix of the reference with its custom op enabled: forward, first-order gradients of the
    image and the grid, and the second-order gradient R1 takes through it (d/d image of |d out / d image|^2)."""
    from torch_utils.ops import grid_sample_gradfix as ref_gs
    g = torch.Generator().manual_seed(105)
    b = Bundle('grid_sample')
    ref_gs.enabled = True
    for n, c, h, w, oh, ow, spread in ((2, 3, 8, 9, 7, 10, 1.2), (1, 4, 16, 16, 20, 12, 0.9), (3, 1, 5, 6, 5, 6, 1.6)):
        x = rnd(g, n, c, h, w).requires_grad_(True)
        grid = ((torch.rand(n, oh, ow, 2, generator=g) * 2 - 1) * spread).requires_grad_(True)
        y = ref_gs.grid_sample(x, grid)
        dy = rnd(g, *y.shape)
        gx, ggrid = torch.autograd.grad((y * dy).sum(), [x, grid], create_graph=True)
        v = rnd(g, *x.shape)
        # second order: how the image gradient reacts to the incoming gradient (linear map: sampler applied to v)
        dy2 = dy.clone().requires_grad_(True)
        gx2, = torch.autograd.grad((ref_gs.grid_sample(x, grid.detach()) * dy2).sum(), x, create_graph=True)
        g_dy, = torch.autograd.grad((gx2 * v).sum(), dy2)
        b.add(dict(op='grid_sample', shape=[n, c, h, w], out=[oh, ow]), x=x, grid=grid, y=y, dy=dy, gx=gx, ggrid=ggrid, v=v, g_dy=g_dy)
    ref_gs.enabled = False
    b.save()


AUGMENT_SPECS = {
    'blit': dict(xflip=1, rotate90=1, xint=1),
    'geom': dict(scale=1, rotate=1, aniso=1, xfrac=1),
    'color': dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1),
    'bgc': dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1,
                saturation=1),
    'filter_cutout': dict(imgfilter=1, cutout=1),
    'bands': dict(imgfilter=1, imgfilter_bands=[1, 0, 1, 0.5], imgfilter_std=0.7),
}


def gen_augment():
    """The reference's AugmentPipe with every random draw replaced by a percentile (its own `debug_percentile`
    switch): outputs, image gradients and - for the geometric part - the gradient of an R1-style penalty."""
    from training import augment as ref_augment
    from torch_utils.ops import grid_sample_gradfix as ref_gs
    ref_gs.enabled = True
    g = torch.Generator().manual_seed(106)
    b = Bundle('augment')
    for name, spec in AUGMENT_SPECS.items():
        pipe = ref_augment.AugmentPipe(**spec)
        for (n, c, h, w) in ((2, 3, 32, 32), (2, 3, 24, 40), (2, 1, 32, 32)):
            if c == 1 and name not in ('color', 'blit'):
                continue
            for q in (0.2, 0.45, 0.8):
                x = (rnd(g, n, c, h, w) * 0.5).requires_grad_(True)
                y = pipe(x, debug_percentile=q)
                dy = rnd(g, *y.shape)
                gx, = torch.autograd.grad((y * dy).sum(), x, create_graph=True)
                extra = {}
                if name in ('geom', 'bgc') and (h, w) == (32, 32):
                    gy, = torch.autograd.grad(pipe(x, debug_percentile=q).square().sum(), x, create_graph=True)
                    r1, = torch.autograd.grad(gy.square().sum(), x)
                    extra = dict(r1=r1)
                b.add(dict(spec=name, kwargs=spec, q=q, shape=[n, c, h, w]), x=x, y=y, dy=dy, gx=gx, **extra)
    b.add(dict(spec='buffers'), Hz_geom=pipe.Hz_geom, Hz_fbank=pipe.Hz_fbank)
    ref_gs.enabled = False
    b.save()


def gen_data_path():
    """The reference's ImageFolderDataset and InfiniteSampler on synthetic PNG folders (written by oracle/data.py, which
    the tests call again to rebuild the same files), and its uint8 -> [-1, 1] conversion of every byte value."""
    import tempfile
    from training import dataset as ref_dataset
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import data as odata
    b = Bundle('data_path')
    with tempfile.TemporaryDirectory() as tmp:
        folders = [  # (name, count, channels, resolution, seed, labels, zip?)
            ('rgb_dir', 11, 3, 16, 21, 'int', False),
            ('rgb_zip', 7, 3, 8, 22, 'float', True),
            ('gray_dir', 5, 1, 16, 23, None, False),
        ]
        for name, count, channels, res, seed, kind, as_zip in folders:
            images = odata.synthetic_images(count, channels, res, seed)
            labels = None
            if kind == 'int':
                labels = [int(v) for v in np.random.RandomState(seed + 100).randint(0, 4, size=count)]
            elif kind == 'float':
                labels = np.random.RandomState(seed + 100).randn(count, 3).astype(np.float32).round(3)
            path = os.path.join(tmp, name + ('.zip' if as_zip else ''))
            if not as_zip:
                os.makedirs(path)
            odata.write_image_folder(path, images, labels, as_zip=as_zip)
            for kwargs in (dict(), dict(use_labels=True), dict(max_size=4, random_seed=5, xflip=True, use_labels=True),
                           dict(xflip=True), dict(max_size=3, random_seed=1)):
                ds = ref_dataset.ImageFolderDataset(path=path, resolution=res, **kwargs)
                items = [ds[i] for i in range(len(ds))]
                b.add(dict(op='dataset', folder=name, count=count, channels=channels, resolution=res, seed=seed, labels=kind,
                           zip=as_zip, kwargs=kwargs, name=ds.name, image_shape=ds.image_shape, label_shape=ds.label_shape,
                           label_dim=ds.label_dim, has_labels=bool(ds.has_labels), has_onehot_labels=bool(ds.has_onehot_labels),
                           length=len(ds), details=[[int(ds.get_details(i).raw_idx), bool(ds.get_details(i).xflip)] for i in range(len(ds))]),
                      images=np.stack([im for im, _ in items]), item_labels=np.stack([lb for _, lb in items]),
                      get_label=np.stack([ds.get_label(i) for i in range(len(ds))]))
                ds.close()
    for size, replicas, shuffle, seed, window in ((10, 1, True, 0, 0.5), (10, 2, True, 3, 0.5), (37, 4, True, 7, 0.5), (10, 1, False, 0, 0.5),
                                                  (10, 3, True, 1, 0.0), (6, 2, True, 2, 1.0), (1, 1, True, 0, 0.5), (3, 1, True, 9, 0.5)):
        class _Sized:
            def __len__(self):
                return size
        per_rank = []
        for rank in range(replicas):
            # (the constructor hands the dataset to torch's Sampler base class, which no longer takes one:
            # the attributes it would set are set here, the iteration is the reference's)
            sampler = ref_misc.InfiniteSampler.__new__(ref_misc.InfiniteSampler)
            sampler.__dict__.update(dataset=_Sized(), rank=rank, num_replicas=replicas, shuffle=shuffle, seed=seed, window_size=window)
            it = iter(sampler)
            per_rank.append([int(next(it)) for _ in range(64)])
        b.add(dict(op='sampler', size=size, num_replicas=replicas, shuffle=shuffle, seed=seed, window_size=window), order=np.asarray(per_rank))
    every = torch.arange(256, dtype=torch.uint8)
    b.add(dict(op='to_float'), u8=every, f32=every.to(torch.float32) / 127.5 - 1)
    b.save()


# ----------------------------------------------------------------------------
# Layers and networks.

def randomize(module, gen, scale=0.2):
    """Make zero-initialised parameters (biases, noise strengths, offsets) matter."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if float(p.abs().max()) == 0:
                p.copy_(torch.randn(p.shape, generator=gen) * scale)


def state_arrays(module, prefix='sd.'):
    return {prefix + k: npy(v) for k, v in module.state_dict().items()}


def gen_layers():
    g = torch.Generator().manual_seed(105)
    b = Bundle('layers')
    combos = [('none', False, 'multiplicative', ('all',)),
              ('out_in_additive', True, 'out_in_additive', ('all',)),
              ('affine_light', True, 'out_in_additive,affine_out_in_5_1_additive', ('synt_weights_offset', 'tRGB_weights_offset')),
              ('additive', True, 'additive', ('all',)),
              ('multiplicative', True, 'multiplicative', ('all',)),
              ('out_in_5_2', True, 'out_in_5_2', ('all',)),
              ('w_space', True, 'multiplicative_w_space', ('all',))]
    for tag, use, par, parts in combos:
        for kind in ('conv_up', 'conv', 'torgb'):
            torch.manual_seed(7)
            with quiet():
                if kind == 'torgb':
                    layer = ref_networks.ToRGBLayer(8, 3, w_dim=16, resolution=16, conv_clamp=256,
                                                    use_domain_modulation=use, domain_modulation_parametrization=par,
                                                    generator_requires_grad_parts=parts)
                else:
                    layer = ref_networks.SynthesisLayer(8, 6, w_dim=16, resolution=16, up=2 if kind == 'conv_up' else 1,
                                                        conv_clamp=256, use_domain_modulation=use,
                                                        domain_modulation_parametrization=par,
                                                        generator_requires_grad_parts=parts)
            randomize(layer, g)
            hin = 8 if kind == 'conv_up' else 16
            x = rnd(g, 2, 8, hin, hin)
            w = rnd(g, 2, 16)
            outs = {}
            for fused in (False, True):
                kw = dict(fused_modconv=fused, noise_mode='const')
                if kind != 'torgb':
                    kw['gain'] = math.sqrt(0.5) if fused else 1.0
                outs['y_fused' if fused else 'y'] = layer(x, w, **kw)
            params = [p for p in layer.parameters()]
            dy = rnd(g, *outs['y'].shape)
            grads = torch.autograd.grad((outs['y'] * dy).sum(), params, allow_unused=True)
            garr = {'g.' + n: (gr if gr is not None else torch.zeros_like(p))
                    for (n, p), gr in zip(layer.named_parameters(), grads)}
            b.add(dict(tag=tag, kind=kind, use_domain_modulation=use, parametrization=par, parts=list(parts)),
                  x=x, w=w, dy=dy, **outs, **state_arrays(layer), **garr)
    b.save()


G_SMALL = dict(z_dim=32, c_dim=0, w_dim=32, img_channels=3, mapping_kwargs=dict(num_layers=2))


def make_G(res, par=None, parts=('all',), channel_base=256, channel_max=16, conv_clamp=256, seed=11, **extra):
    torch.manual_seed(seed)
    sk = dict(channel_base=channel_base, channel_max=channel_max, num_fp16_res=0, conv_clamp=conv_clamp,
              generator_requires_grad_parts=list(parts))
    if par is not None:
        sk.update(use_domain_modulation=True, domain_modulation_parametrization=par)
    kw = dict(G_SMALL)
    kw.update(extra)
    with quiet():
        return ref_networks.Generator(img_resolution=res, synthesis_kwargs=sk, **kw)


def make_D(res, channel_base=256, channel_max=16, conv_clamp=256, mbstd=4, seed=12, **extra):
    torch.manual_seed(seed)
    with quiet():
        return ref_networks.Discriminator(c_dim=extra.pop('c_dim', 0), img_resolution=res, img_channels=3,
                                          channel_base=channel_base, channel_max=channel_max, num_fp16_res=0,
                                          conv_clamp=conv_clamp, epilogue_kwargs=dict(mbstd_group_size=mbstd), **extra)


AFFINE_PLUS = ['synt_affine', 'tRGB_affine', 'synt_weights_offset.b16', 'tRGB_weights_offset.b16']
AFFINE_LIGHT = ['synt_affine_weights_offset', 'tRGB_affine_weights_offset', 'synt_weights_offset.b16',
                'tRGB_weights_offset.b16']


def gen_networks():
    g = torch.Generator().manual_seed(106)
    b = Bundle('networks')
    gcases = [('G32_full', 32, None, ('all',), {}),
              ('G32_affine_plus', 32, 'out_in_additive', AFFINE_PLUS, {}),
              ('G64_affine_light', 64, 'out_in_additive,affine_out_in_5_1_additive', AFFINE_LIGHT, {}),
              ('G32_stylespace', 32, 'additive', ('synt_offset', 'tRGB_offset'), {}),
              ('G16_cond', 16, None, ('all',), dict(c_dim=5))]
    for tag, res, par, parts, extra in gcases:
        G = make_G(res, par, parts, **extra)
        randomize(G, g)
        G.train()
        ref_training_loop.set_requires_grad(G, list(parts))
        trainable = [n for n, p in G.named_parameters() if p.requires_grad]
        z = rnd(g, 4, 32)
        c = torch.zeros(4, 0) if G.c_dim == 0 else torch.eye(G.c_dim)[[0, 2, 4, 1]]
        G.mapping.w_avg.copy_(rnd(g, 32) * 0.1)
        sd = state_arrays(G)  # before the forward pass updates w_avg
        img = G(z, c, noise_mode='const')
        w_avg_after = G.mapping.w_avg.clone()
        dy = rnd(g, *img.shape)
        grads = torch.autograd.grad((img * dy).sum(), [p for p in G.parameters() if p.requires_grad])
        garr = {'g.' + n: gr for n, gr in zip(trainable, grads)}
        G.eval()
        img_eval = G(z, c, noise_mode='const', truncation_psi=0.7, truncation_cutoff=3)   # fused path, truncation
        img_none = G(z, c, noise_mode='none')
        b.add(dict(tag=tag, kind='G', res=res, parametrization=par, parts=list(parts), c_dim=G.c_dim,
                   trainable=trainable, channel_base=256, channel_max=16, conv_clamp=256, mapping_layers=2, z_dim=32, w_dim=32),
              z=z, c=c, img=img, img_eval=img_eval, img_none=img_none, dy=dy, w_avg_after=w_avg_after, **sd, **garr)

    for tag, res, extra, n in [('D32', 32, {}, 4), ('D64', 64, {}, 8), ('D16_cond', 16, dict(c_dim=5), 4),
                               ('D32_skip', 32, dict(architecture='skip'), 4),
                               ('D32_freeze', 32, dict(block_kwargs=dict(freeze_layers=3)), 4)]:
        D = make_D(res, **extra)
        randomize(D, g)
        D.train().requires_grad_(True)
        img = rnd(g, n, 3, res, res).requires_grad_(True)
        c = torch.zeros(n, 0) if D.c_dim == 0 else torch.eye(D.c_dim)[torch.arange(n) % D.c_dim]
        logits = D(img, c)
        names = [k for k, _ in D.named_parameters()]
        grads = torch.autograd.grad(logits.sum(), [img] + list(D.parameters()), create_graph=True)
        # R1-style second order: d/dparams of |d logits / d img|^2
        r1 = grads[0].square().sum()
        g2 = torch.autograd.grad(r1, list(D.parameters()), allow_unused=True)
        garr = {'g.' + k: gr for k, gr in zip(names, grads[1:])}
        g2arr = {'g2.' + k: (gr if gr is not None else torch.zeros_like(p))
                 for (k, p), gr in zip(D.named_parameters(), g2)}
        b.add(dict(tag=tag, kind='D', res=res, c_dim=D.c_dim, architecture=extra.get('architecture', 'resnet'),
                   channel_base=256, channel_max=16, conv_clamp=256, mbstd_group_size=4),
              img=img, c=c, logits=logits, gimg=grads[0], r1=r1, **state_arrays(D), **garr, **g2arr)
    b.save()


class ConstNoiseSynthesis(torch.nn.Module):
    """Synthesis network called with deterministic (constant) noise inputs."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, ws):
        return self.net(ws, noise_mode='const')


def gen_loss():
    g = torch.Generator().manual_seed(107)
    b = Bundle('loss')
    for tag, par, parts in [('affine_plus', 'out_in_additive', AFFINE_PLUS), ('full', None, ['all'])]:
        G = make_G(32, par, parts)
        D = make_D(32)
        randomize(G, g)
        randomize(D, g)
        G.train().requires_grad_(False)
        D.train().requires_grad_(False)
        sdG, sdD = state_arrays(G, 'G.'), state_arrays(D, 'D.')
        loss = ref_loss.StyleGAN2Loss(device=torch.device('cpu'), G_mapping=G.mapping,
                                      G_synthesis=ConstNoiseSynthesis(G.synthesis), D=D, augment_pipe=None,
                                      style_mixing_prob=0, r1_gamma=10, pl_batch_shrink=2, pl_decay=0.01, pl_weight=2)
        z = rnd(g, 4, 32)
        c = torch.zeros(4, 0)
        real = (torch.randint(0, 256, [4, 3, 32, 32], generator=g).to(torch.float32) / 127.5 - 1)
        phase_out = {}
        for phase, module, pparts, gain in [('Gmain', G, parts, 1), ('Greg', G, parts, 4),
                                            ('Dmain', D, ['all'], 1), ('Dreg', D, ['all'], 16)]:
            G.zero_grad(set_to_none=True)
            D.zero_grad(set_to_none=True)
            G.requires_grad_(False)
            D.requires_grad_(False)
            ref_training_loop.set_requires_grad(module, list(pparts))
            w_avg0 = G.mapping.w_avg.clone()
            pl_mean0 = loss.pl_mean.clone()
            torch.manual_seed(1234)   # the only draw is the path-length noise (Greg)
            loss.accumulate_gradients(phase=phase, real_img=real, real_c=c, gen_z=z, gen_c=c, sync=True, gain=gain)
            for n, p in module.named_parameters():
                if p.grad is not None:
                    phase_out[f'{phase}.g.{n}'] = p.grad.clone()
            phase_out[f'{phase}.w_avg_before'] = w_avg0
            phase_out[f'{phase}.w_avg_after'] = G.mapping.w_avg.clone()
            phase_out[f'{phase}.pl_mean_before'] = pl_mean0
            phase_out[f'{phase}.pl_mean_after'] = loss.pl_mean.clone()
            module.requires_grad_(False)
        torch.manual_seed(1234)
        pl_noise_unit = torch.randn(2, 3, 32, 32)   # same stream as torch.randn_like(gen_img) in Greg
        b.add(dict(tag=tag, parametrization=par, parts=list(parts), res=32, style_mixing_prob=0, r1_gamma=10,
                   pl_weight=2, pl_decay=0.01, pl_batch_shrink=2, gains=dict(Gmain=1, Greg=4, Dmain=1, Dreg=16),
                   pl_noise_seed=1234, channel_base=256, channel_max=16, conv_clamp=256, mapping_layers=2,
                   z_dim=32, w_dim=32, mbstd_group_size=4),
              z=z, c=c, real=real, pl_noise_unit=pl_noise_unit, **sdG, **sdD, **phase_out)
    b.save()


def seeded_fill(name, shape):
    """Deterministic parameter values from the tensor's name (crc32 seed)."""
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7fffffff)
    t = torch.randn(list(shape), generator=gen)
    if name.endswith('noise_strength'):
        t = t * 0.1
    elif name.endswith('.bias') and '.affine.' in name:
        t = t * 0.1 + 1
    elif name.endswith('.bias') or 'weights_offset' in name or name.endswith('w_avg'):
        t = t * 0.1
    return t


def gen_full_size():
    """FFHQ-256 generator (paper256 channels, Affine+), batch 1: image checksum + crop.
    Parameters come from `seeded_fill`, so the test rebuilds them instead of storing 100 MB."""
    b = Bundle('full_size')
    torch.manual_seed(0)
    with quiet():
        G = ref_networks.Generator(
            z_dim=512, c_dim=0, w_dim=512, img_resolution=256, img_channels=3, mapping_kwargs=dict(num_layers=8),
            synthesis_kwargs=dict(channel_base=16384, channel_max=512, num_fp16_res=0, conv_clamp=256,
                                  use_domain_modulation=True, domain_modulation_parametrization='out_in_additive',
                                  generator_requires_grad_parts=['synt_affine', 'tRGB_affine',
                                                                 'synt_weights_offset.b64', 'tRGB_weights_offset.b64']))
    names = []
    with torch.no_grad():
        for name, t in list(G.named_parameters()) + list(G.named_buffers()):
            if name.endswith('resample_filter') or name.endswith('ones_weights') or name.endswith('offset_mask'):
                continue
            t.copy_(seeded_fill(name, t.shape))
            names.append(name)
    G.train()
    z = torch.randn(1, 512, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        img = G(z, torch.zeros(1, 0), noise_mode='const')
    b.add(dict(names=names, shapes=[list(dict(list(G.named_parameters()) + list(G.named_buffers()))[n].shape) for n in names],
               res=256, channel_base=16384, parametrization='out_in_additive', z_seed=5),
          z=z, crop=img[:, :, 120:136, 120:136], sum=img.double().sum(), abs_sum=img.double().abs().sum(),
          row_sums=img.double().sum(dim=[0, 1, 3]))
    b.save()


def _fill_all(module):
    """`seeded_fill` into every parameter / buffer that the state dict carries a value for; returns the names."""
    names = []
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if name.endswith('resample_filter') or name.endswith('ones_weights') or name.endswith('offset_mask'):
                continue
            t.copy_(seeded_fill(name, t.shape))
            names.append(name)
    return names


def _shapes(module, names):
    every = dict(list(module.named_parameters()) + list(module.named_buffers()))
    return [list(every[n].shape) for n in names]


def _summary(prefix, t, crop=None):
    """What the fixtures keep of a large tensor: a crop, the sums over everything but the rows, and two checksums."""
    t = t.detach()
    out = {prefix + 'sum': t.double().sum(), prefix + 'abs_sum': t.double().abs().sum(),
           prefix + 'row_sums': t.double().sum(dim=[0, 1, 3])}
    if crop is not None:
        out[prefix + 'crop'] = t[:, :, crop[0]:crop[1], crop[0]:crop[1]]
    return out


def _grad_entries(name, grad):
    """A gradient whole when it is small, else every 61st element of it plus its L2 norm."""
    if grad.numel() <= 65536:
        return {'g.' + name: grad}
    return {'gs.' + name: grad.flatten()[::61], 'gnorm.' + name: grad.double().norm()}


def _exact_generator_gradients(G, names, trainable, z, dy, res, channel_base, par):
    """The same gradients in float64, from the functional restatement of the networks under oracle/ (the reference's
    blocks cast to float32 themselves, networks.py:1031-1034, so it cannot run in float64).  Sums of 1e5 and more
    signed terms that nearly cancel make these gradients ill-conditioned in float32: the distance of the reference's
    own float32 result from the float64 one is what the tests scale their tolerance with."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import stylegan2 as osg
    every = dict(list(G.named_parameters()) + list(G.named_buffers()))
    p = {n: every[n].detach().double() for n in names}
    for k in trainable:
        p[k].requires_grad_()
    cfg = osg.GeneratorConfig(img_resolution=res, channel_base=channel_base, conv_clamp=256, use_domain_modulation=par is not None,
                              domain_modulation_parametrization=par or 'multiplicative')
    img = osg.generator(p, z.double(), torch.zeros(1, 0, dtype=torch.float64), cfg, noise_mode='const', fused_modconv=False)
    return [g.float() for g in torch.autograd.grad((img * dy.double()).sum(), [p[k] for k in trainable])]


def _add_generator_case(b, tag, res, channel_base, par, parts, z_seed):
    """One full-size generator case of gen_full_size_grads / gen_full_size_grads_more.  par=None: no domain modulation."""
    torch.manual_seed(0)
    use_dm = par is not None
    with quiet():
        G = ref_networks.Generator(
            z_dim=512, c_dim=0, w_dim=512, img_resolution=res, img_channels=3, mapping_kwargs=dict(num_layers=8),
            synthesis_kwargs=dict(channel_base=channel_base, channel_max=512, num_fp16_res=0, conv_clamp=256,
                                  use_domain_modulation=use_dm, domain_modulation_parametrization=par if use_dm else 'multiplicative',
                                  generator_requires_grad_parts=parts))
    names = _fill_all(G)
    G.train()
    ref_training_loop.set_requires_grad(G, list(parts))
    trainable = [n for n, p in G.named_parameters() if p.requires_grad]
    z = torch.randn(1, 512, generator=torch.Generator().manual_seed(z_seed))
    img = G(z, torch.zeros(1, 0), noise_mode='const')
    dy = torch.randn(img.shape, generator=torch.Generator().manual_seed(z_seed + 100))
    grads = torch.autograd.grad((img * dy).sum(), [p for p in G.parameters() if p.requires_grad])
    mid = res // 2
    exact = _exact_generator_gradients(G, names, trainable, z, dy, res, channel_base, par)
    b.add(dict(tag=tag, kind='G', res=res, channel_base=channel_base, parametrization=par if use_dm else '', use_domain_modulation=use_dm,
               parts=parts, names=names, shapes=_shapes(G, names), trainable=trainable, z_seed=z_seed, dy_seed=z_seed + 100),
          **_summary('img_', img, (mid - 8, mid + 8)),
          **{k: v for n, gr in zip(trainable, grads) for k, v in _grad_entries(n, gr).items()},
          **{'x' + k: v for n, gr in zip(trainable, exact) for k, v in _grad_entries(n, gr).items() if not k.startswith('gnorm.')})
    print(tag, 'trainable tensors', len(trainable), 'elements', sum(int(gr.numel()) for gr in grads), flush=True)


def gen_full_size_grads_more():
    """Two more full-size generator cases in the layout of gen_full_size_grads (a bundle of their own, so that the
    first one keeps its bytes): the 1024 x 1024 generator with EVERY parameter trainable (full fine-tune, no domain
    modulation) and the 512 x 512 Affine+ generator."""
    b = Bundle('full_size_grads_more')
    _add_generator_case(b, 'G1024_full', 1024, 32768, None, ['all'], 7)
    _add_generator_case(b, 'G512_affine_plus', 512, 32768, 'out_in_additive',
                        ['synt_affine', 'tRGB_affine', 'synt_weights_offset.b64', 'tRGB_weights_offset.b64'], 8)
    b.save()


def gen_full_size_grads():
    """Full-size networks in float32 on seeded parameters and seeded inputs (the tests rebuild both):
    * generators at 256 (Affine+) and 1024 (AffineLight+), batch 1, training mode: image summary and the gradient of
      sum(img * dy) with respect to every trainable parameter (large ones as a strided sample plus their norm), from
      the reference in float32 ('g.' / 'gs.') and in float64 arithmetic ('xg.' / 'xgs.');
    * discriminators at 256 / 512 / 1024: logits, a summary of the image gradient of sum(logits), every bias gradient
      and the L2 norm of every weight gradient."""
    b = Bundle('full_size_grads')
    gcases = [('G256_affine_plus', 256, 16384, 'out_in_additive',
               ['synt_affine', 'tRGB_affine', 'synt_weights_offset.b64', 'tRGB_weights_offset.b64'], 5),
              ('G1024_affine_light', 1024, 32768, 'out_in_additive,affine_out_in_5_1_additive',
               ['synt_affine_weights_offset', 'tRGB_affine_weights_offset', 'synt_weights_offset.b64',
                'tRGB_weights_offset.b64'], 6)]
    for tag, res, channel_base, par, parts, z_seed in gcases:
        _add_generator_case(b, tag, res, channel_base, par, parts, z_seed)

    for tag, res, channel_base, mbstd, n, seed in [('D256', 256, 16384, 8, 4, 21), ('D512', 512, 32768, 8, 2, 22),
                                                   ('D1024', 1024, 32768, 4, 2, 23)]:
        torch.manual_seed(0)
        with quiet():
            D = ref_networks.Discriminator(c_dim=0, img_resolution=res, img_channels=3, channel_base=channel_base,
                                           channel_max=512, num_fp16_res=0, conv_clamp=256,
                                           epilogue_kwargs=dict(mbstd_group_size=mbstd))
        names = _fill_all(D)
        D.train().requires_grad_(True)
        img = torch.randn(n, 3, res, res, generator=torch.Generator().manual_seed(seed)).clamp_(-1, 1).requires_grad_(True)
        logits = D(img, torch.zeros(n, 0))
        pnames = [k for k, _ in D.named_parameters()]
        grads = torch.autograd.grad(logits.sum(), [img] + list(D.parameters()))
        gb = {'g.' + k: gr for k, gr in zip(pnames, grads[1:]) if k.endswith('.bias')}
        gn = {'gnorm.' + k: gr.double().norm() for k, gr in zip(pnames, grads[1:]) if not k.endswith('.bias')}
        mid = res // 2
        b.add(dict(tag=tag, kind='D', res=res, channel_base=channel_base, mbstd_group_size=mbstd, batch=n, img_seed=seed,
                   names=names, shapes=_shapes(D, names)),
              logits=logits, **_summary('gimg_', grads[0], (mid - 8, mid + 8)), **gb, **gn)
        print(tag, 'logits', logits.flatten().tolist(), flush=True)
        del D, img, grads
    b.save()


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['upfirdn2d', 'bias_act', 'conv2d_resample', 'modulated_conv2d', 'misc_ops', 'grid_sample', 'augment', 'data_path', 'layers',
                             'networks', 'loss', 'full_size', 'full_size_grads']
    for name in which:
        globals()['gen_' + name]()
