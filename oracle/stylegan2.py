"""Oracle (test infrastructure): StyleGAN2 generator / discriminator on CPU as
pure functions over a parameter dictionary.

`params` maps the reference's `state_dict()` names to tensors, so a state dict
of the reference's `Generator` / `Discriminator` drives these functions
unchanged.  Restates:
  * normalize_2nd_moment, modulated_conv2d          training/networks.py:584-668
  * w_to_s, weight_to_weight (domain modulation)    training/networks.py:474-579
  * FullyConnectedLayer, Conv2dLayer                training/networks.py:673-760
  * MappingNetwork                                  training/networks.py:765-842
  * SynthesisLayer, ToRGBLayer                      training/networks.py:847-963
  * SynthesisBlock, SynthesisNetwork, Generator     training/networks.py:968-1171
    (the image skip is up-sampled only when an image exists, i.e. the upstream
    block wiring; the fork's lines 1057-1063 crash on the first block)
  * DiscriminatorBlock, Epilogue, Discriminator     training/networks.py:1176-1437
All arithmetic runs in the dtype of the inputs (float32 or float64); the
half-precision pre-normalisation of training/networks.py:621-627 is a no-op
for these dtypes and is not restated.

Inside `low_precision(dtype, from_resolution)` the blocks at and above that
resolution (the reference's `use_fp16` blocks, networks.py:1031, :1245) still
compute in the inputs' dtype, but every tensor that an implementation with
16-bit storage keeps in memory there is rounded to `dtype` and back: layer
inputs after modulation, weights and biases, layer outputs, the tensor between
a resampling filter and its convolution, and - in the backward pass - the
gradients with respect to layer inputs and convolution outputs.  Sums inside
a convolution or filter stay unrounded (16-bit kernels accumulate in float32).
This is the yardstick for 16-bit runs: how far rounding alone moves a result.
"""
import contextlib
import math
import re
from dataclasses import dataclass, field
from typing import Optional, Sequence

import torch

from . import pointwise
from . import resample

_STYLE_KINDS = ('multiplicative', 'additive', 'multiplicative_w_space', 'additive_w_space')
_AFFINE_RE = re.compile(r'^affine_out_in_([0-9]+)_([0-9]+)(?:_additive)?$')
_WEIGHT_RE = re.compile(
    r'^(?:in|out|spatial|in_spatial|out_spatial|out_in|out\+in|out_in_[0-9]+|out_in_[0-9]+_dual|'
    r'out_in_[0-9]+_[0-9]+|out_in_[0-9]+_[0-9]+_train_in|out_in_[0-9]+_[0-9]+_train_out)(?:_additive)?$')


def split_parametrization(use_domain_modulation, spec):
    """'a,b,c' -> (style-space kind, conv-weight kind, affine-weight kind);
    at most one of each (training/networks.py:87-137)."""
    if not use_domain_modulation or not spec:
        return None, None, None
    style, weights, affine = [], [], []
    for name in spec.split(','):
        if _AFFINE_RE.match(name):
            affine.append(name)
        elif _WEIGHT_RE.match(name):
            weights.append(name)
        else:
            style.append(name)
    for group in (style, weights, affine):
        if len(group) > 1:
            raise ValueError(f'Too many types of domain modulation: {group}')
    if style and style[0] not in _STYLE_KINDS:
        raise ValueError(f'Unknown type of domain modulation: {style[0]}')
    return (style[0] if style else None, weights[0] if weights else None, affine[0] if affine else None)


@dataclass
class GeneratorConfig:
    img_resolution: int
    z_dim: int = 512
    c_dim: int = 0
    w_dim: int = 512
    img_channels: int = 3
    channel_base: int = 32768
    channel_max: int = 512
    conv_clamp: Optional[float] = None
    architecture: str = 'skip'
    mapping_layers: int = 8
    mapping_lr_multiplier: float = 0.01
    w_avg_beta: Optional[float] = 0.995
    use_noise: bool = True
    use_domain_modulation: bool = False
    domain_modulation_parametrization: str = 'multiplicative'

    @property
    def block_resolutions(self):
        return [2 ** i for i in range(2, int(math.log2(self.img_resolution)) + 1)]

    def channels(self, res):
        return min(self.channel_base // res, self.channel_max)

    @property
    def num_ws(self):
        n = 0
        for res in self.block_resolutions:
            n += 1 if res == 4 else 2
        return n + 1  # the last block's ToRGB


@dataclass
class DiscriminatorConfig:
    img_resolution: int
    c_dim: int = 0
    img_channels: int = 3
    architecture: str = 'resnet'
    channel_base: int = 32768
    channel_max: int = 512
    conv_clamp: Optional[float] = None
    cmap_dim: Optional[int] = None
    mbstd_group_size: Optional[int] = 4
    mbstd_num_channels: int = 1
    mapping_layers: int = 8

    @property
    def block_resolutions(self):
        return [2 ** i for i in range(int(math.log2(self.img_resolution)), 2, -1)]

    def channels(self, res):
        return min(self.channel_base // res, self.channel_max)


_FILTER = resample.make_filter([1, 3, 3, 1])


# ----------------------------------------------------------------------------
# 16-bit storage emulation (see the module text).

_LOW = None      # (dtype, first resolution, channels above which an up-sampling layer runs as two 16-bit passes)


@contextlib.contextmanager
def low_precision(dtype=torch.bfloat16, from_resolution=8, two_pass_up_above=128):
    global _LOW
    old, _LOW = _LOW, (dtype, from_resolution, two_pass_up_above)
    try:
        yield
    finally:
        _LOW = old


def _low_dtype(res):
    return _LOW[0] if _LOW is not None and res >= _LOW[1] else None


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype, values, gradients):
        ctx.dtype, ctx.gradients = dtype, gradients
        return x.to(dtype).to(x.dtype) if values else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dtype).to(g.dtype) if ctx.gradients else g), None, None, None


class _LowEpilogue(torch.autograd.Function):
    """y = clamp(lrelu(v * scale[n, c] + noise + bias[c]) * gain) stored in 16 bits, with the backward pass of an
    implementation that keeps nothing but that stored output: slope and clamp mask from the sign and size of y, the
    pre-activation recovered from y for the gradient of `scale`, and the gradient with respect to v rounded to 16
    bits.  (First order only.)"""

    @staticmethod
    def forward(ctx, v, scale, noise, bias, alpha, gain, clamp, low):
        u = v * scale.to(v.dtype).reshape(v.shape[0], -1, 1, 1) + bias.to(v.dtype).reshape(1, -1, 1, 1)
        if noise is not None:
            u = u + noise.to(v.dtype)
        y = torch.where(u > 0, u, u * alpha) * gain
        if clamp is not None:
            y = y.clamp(-clamp, clamp)
        y = y.to(low).to(v.dtype)
        ctx.save_for_backward(y, scale, noise, bias)
        ctx.cfg = (alpha, gain, clamp, low)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale, noise, bias = ctx.saved_tensors
        alpha, gain, clamp, low = ctx.cfg
        n = y.shape[0]
        g = dy * gain
        g = torch.where(y > 0, g, g * alpha)
        if clamp is not None:
            g = torch.where((y > -clamp) & (y < clamp), g, torch.zeros_like(g))
        sc = scale.to(y.dtype).reshape(n, -1, 1, 1)
        dv = (g * sc).to(low).to(y.dtype)
        t = y / gain
        t = torch.where(t < 0, t / alpha, t)                       # the pre-activation, from the stored output
        rest = t - bias.to(y.dtype).reshape(1, -1, 1, 1)
        if noise is not None:
            rest = rest - noise.to(y.dtype)
        dscale = ((g * rest).sum(dim=[2, 3]) / scale.to(y.dtype)).to(scale.dtype)
        dnoise = None
        if noise is not None and noise.requires_grad:
            dnoise = g.sum(dim=1, keepdim=True)
            while dnoise.ndim > noise.ndim:
                dnoise = dnoise.sum(dim=0)
            for d in range(noise.ndim):
                if noise.shape[d] == 1 and dnoise.shape[d] != 1:
                    dnoise = dnoise.sum(dim=d, keepdim=True)
            dnoise = dnoise.to(noise.dtype)
        dbias = g.sum(dim=[0, 2, 3]).to(bias.dtype)
        return dv, dscale, dnoise, dbias, None, None, None, None


def _rv(x, low):
    """Round the values of x to the 16-bit dtype `low` (None: leave)."""
    return x if low is None or x is None else _Round.apply(x, low, True, False)


def _rg(x, low):
    """Round the gradient that flows back through this point."""
    return x if low is None or x is None or not x.requires_grad else _Round.apply(x, low, False, True)


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


# ----------------------------------------------------------------------------
# Domain-modulation helpers.

def _low_rank_terms(p, prefix):
    terms = []
    k = 0
    while f'{prefix}weights_offset_in_{k}' in p:
        terms.append(p[f'{prefix}weights_offset_out_{k}'] @ p[f'{prefix}weights_offset_in_{k}'])
        k += 1
    return terms


def effective_affine_weight(p, prefix, kind):
    """Affine (fully connected) weight with its low-rank offset; the terms are
    summed (training/networks.py:545-562)."""
    w = p[prefix + 'weight']
    if kind is None:
        return w
    off = sum(_low_rank_terms(p, prefix))
    return w + off if kind.endswith('additive') else (1 + off) * w


def effective_conv_weight(p, prefix, kind):
    """Conv weight with its offset (training/networks.py:564-579): the
    decomposed forms average their rank-k terms, the base forms broadcast."""
    w = p[prefix + 'weight']
    if kind is None:
        return w
    if prefix + 'weights_offset' in p:
        off = p[prefix + 'weights_offset']
    elif kind.startswith('out+in'):
        off = (p[prefix + 'weights_offset_out_0'] + p[prefix + 'weights_offset_in_0'])[:, :, None, None]
    elif prefix + 'weights_offset_in_0' in p:
        terms = _low_rank_terms(p, prefix)
        off = sum(terms)[:, :, None, None] / len(terms)
    else:
        return w  # this layer was not selected for a weight offset
    return w + off if kind.endswith('additive') else (1 + off) * w


def fully_connected(p, prefix, x, activation='linear', lr_multiplier=1.0, affine_kind=None):
    w = effective_affine_weight(p, prefix, affine_kind)
    w = w.to(x.dtype) * (lr_multiplier / math.sqrt(w.shape[1]))
    b = p.get(prefix + 'bias')
    if b is not None:
        b = b.to(x.dtype)
        if lr_multiplier != 1:
            b = b * lr_multiplier
    if activation == 'linear' and b is not None:
        return torch.addmm(b.unsqueeze(0), x, w.t())
    return pointwise.bias_act(x.matmul(w.t()), b, act=activation)


def styles_from_w(p, prefix, w, weight_gain, kinds):
    """W -> S with the optional W-space / S-space offsets (w_to_s)."""
    style_kind, _, affine_kind = kinds
    mask = p.get(prefix + 'offset_mask', 1.0)
    if style_kind == 'multiplicative_w_space':
        w = (1 + p[prefix + 'offset'] * mask) * w
    elif style_kind == 'additive_w_space':
        w = w + p[prefix + 'offset'] * mask
    s = fully_connected(p, prefix + 'affine.', w, affine_kind=affine_kind) * weight_gain
    if style_kind == 'multiplicative':
        s = (1 + p[prefix + 'offset'] * mask) * s
    elif style_kind == 'additive':
        s = s + p[prefix + 'offset'] * mask
    return s


# ----------------------------------------------------------------------------
# Layers.

def modulated_conv2d(x, weight, styles, noise=None, up=1, padding=0, resample_filter=None,
                     demodulate=True, flip_weight=True, fused_modconv=True, low=None, epilogue=None):
    n = x.shape[0]
    cout, cin, kh, kw = weight.shape
    w = dcoefs = None
    if demodulate or fused_modconv:
        w = weight.unsqueeze(0) * styles.reshape(n, 1, -1, 1, 1)
    if demodulate:
        dcoefs = (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
    if not fused_modconv:
        # scale activations, convolve with the shared weight, rescale outputs
        x = _rg(x, low)
        if low is not None and kh == 1:       # (16-bit 1x1 layers fold the styles into their few weights in float32)
            x = x * styles.to(x.dtype).reshape(n, -1, 1, 1)
        else:
            x = _rv(x * styles.to(x.dtype).reshape(n, -1, 1, 1), low)
        # an up-sampling layer keeps its transposed-convolution output in 16 bits when it runs as two passes (wide
        # layers); its backward pass always has the gradient of that tensor in 16 bits
        two_pass = low is not None and up > 1 and cin > _LOW[2]
        x = resample.conv2d_resample(x, _rv(weight.to(x.dtype), low), f=resample_filter, up=up, padding=padding,
                                     flip_weight=flip_weight, between=lambda t: _rg(_rv(t, low) if two_pass else t, low))
        if low is not None and demodulate and epilogue is not None:
            # the layer's epilogue as ONE operation on the convolution output (see _LowEpilogue)
            bias, alpha, gain, clamp = epilogue
            return _LowEpilogue.apply(x, dcoefs, noise, bias, alpha, gain, clamp, low)
        x = _rg(x, low)
        if demodulate:
            x = x * dcoefs.to(x.dtype).reshape(n, -1, 1, 1)
        if noise is not None:
            x = x + noise.to(x.dtype)
        return x
    # one grouped convolution with per-sample weights
    if demodulate:
        w = w * dcoefs.reshape(n, -1, 1, 1, 1)
    x = x.reshape(1, -1, *x.shape[2:])
    x = resample.conv2d_resample(x, w.reshape(-1, cin, kh, kw).to(x.dtype), f=resample_filter, up=up,
                                 padding=padding, groups=n, flip_weight=flip_weight)
    x = x.reshape(n, -1, *x.shape[2:])
    if noise is not None:
        x = x + noise
    return x


def conv2d_layer(p, prefix, x, activation='linear', up=1, down=1, conv_clamp=None, gain=1.0, low=None):
    w = p[prefix + 'weight']
    cout, cin, k, _ = w.shape
    w = _rv((w * (1 / math.sqrt(cin * k * k))).to(x.dtype), low)
    b = p.get(prefix + 'bias')
    b = _rv(b.to(x.dtype), low) if b is not None else None
    x = resample.conv2d_resample(_rg(x, low), w, f=_FILTER, up=up, down=down, padding=k // 2, flip_weight=(up == 1),
                                 between=lambda t: _rg(_rv(t, low), low))
    act_gain = pointwise.ACTIVATIONS[activation][2] * gain
    act_clamp = conv_clamp * gain if conv_clamp is not None else None
    return _rv(pointwise.bias_act(_rg(x, low), b, act=activation, gain=act_gain, clamp=act_clamp), low)


def synthesis_layer(p, prefix, x, w, cfg, kinds, up=1, noise_mode='const', noise=None, fused_modconv=False, gain=1.0, low=None):
    styles = styles_from_w(p, prefix, w, 1.0, kinds)
    weight = effective_conv_weight(p, prefix, kinds[1])
    res = x.shape[2] * up
    nz = None
    if cfg.use_noise and noise_mode == 'random':
        rnd = noise if noise is not None else torch.randn([x.shape[0], 1, res, res], dtype=x.dtype)
        nz = rnd * p[prefix + 'noise_strength']
    elif cfg.use_noise and noise_mode == 'const':
        nz = p[prefix + 'noise_const'] * p[prefix + 'noise_strength']
    act_gain = pointwise.ACTIVATIONS['lrelu'][2] * gain
    act_clamp = cfg.conv_clamp * gain if cfg.conv_clamp is not None else None
    if low is not None and not fused_modconv:
        return modulated_conv2d(x, weight, styles, noise=nz, up=up, padding=weight.shape[2] // 2, resample_filter=_FILTER,
                                flip_weight=(up == 1), fused_modconv=False, low=low,
                                epilogue=(_rv(p[prefix + 'bias'].to(x.dtype), low), pointwise.ACTIVATIONS['lrelu'][1], act_gain, act_clamp))
    x = modulated_conv2d(x, weight, styles, noise=nz, up=up, padding=weight.shape[2] // 2, resample_filter=_FILTER,
                         flip_weight=(up == 1), fused_modconv=fused_modconv, low=low)
    return _rv(pointwise.bias_act(x, _rv(p[prefix + 'bias'].to(x.dtype), low), act='lrelu', gain=act_gain, clamp=act_clamp), low)


def torgb_layer(p, prefix, x, w, cfg, kinds, fused_modconv=False, low=None):
    weight = effective_conv_weight(p, prefix, kinds[1])
    styles = styles_from_w(p, prefix, w, 1 / math.sqrt(weight.shape[1] * weight.shape[2] ** 2), kinds)
    x = modulated_conv2d(x, weight, styles, demodulate=False, fused_modconv=fused_modconv, low=low)
    return _rg(_rv(pointwise.bias_act(x, _rv(p[prefix + 'bias'].to(x.dtype), low), clamp=cfg.conv_clamp), low), low)


# ----------------------------------------------------------------------------
# Generator.

def mapping(p, z, c, cfg, prefix='mapping.', num_ws='cfg', truncation_psi=1.0, truncation_cutoff=None,
            z_dim=None, c_dim=None, num_layers=None):
    z_dim = cfg.z_dim if z_dim is None else z_dim
    c_dim = cfg.c_dim if c_dim is None else c_dim
    num_layers = cfg.mapping_layers if num_layers is None else num_layers
    num_ws = cfg.num_ws if num_ws == 'cfg' else num_ws
    x = None
    if z_dim > 0:
        x = normalize_2nd_moment(z)
    if c_dim > 0:
        y = normalize_2nd_moment(fully_connected(p, prefix + 'embed.', c))
        x = torch.cat([x, y], dim=1) if x is not None else y
    lrm = getattr(cfg, 'mapping_lr_multiplier', 0.01)
    for i in range(num_layers):
        x = fully_connected(p, f'{prefix}fc{i}.', x, activation='lrelu', lr_multiplier=lrm)
    if num_ws is not None:
        x = x.unsqueeze(1).repeat([1, num_ws, 1])
    if truncation_psi != 1:
        w_avg = p[prefix + 'w_avg']
        if num_ws is None or truncation_cutoff is None:
            x = w_avg.lerp(x, truncation_psi)
        else:
            x = torch.cat([w_avg.lerp(x[:, :truncation_cutoff], truncation_psi), x[:, truncation_cutoff:]], dim=1)
    return x


def w_avg_update(w_avg, w_batch, beta):
    """Moving average of W tracked in training mode (networks.py:825-827)."""
    return w_batch.detach().mean(dim=0).lerp(w_avg, beta)


def synthesis(p, ws, cfg, prefix='synthesis.', noise_mode='const', fused_modconv=False, noises=None, taps=None):
    kinds = split_parametrization(cfg.use_domain_modulation, cfg.domain_modulation_parametrization)
    ws = ws.to(torch.float64 if ws.dtype == torch.float64 else torch.float32)
    x = img = None
    w_idx = 0
    for res in cfg.block_resolutions:
        bp = f'{prefix}b{res}.'
        num_conv = 1 if res == 4 else 2
        has_rgb = res == cfg.img_resolution or cfg.architecture == 'skip'
        bw = ws[:, w_idx:w_idx + num_conv + (1 if has_rgb else 0)]
        w_idx += num_conv
        it = iter(bw.unbind(dim=1))
        nz = (noises or {}).get
        low = _low_dtype(res)
        lk = dict(cfg=cfg, kinds=kinds, noise_mode=noise_mode, fused_modconv=fused_modconv, low=low)
        if x is not None:
            x = _rv(x, low)     # (entering the first 16-bit block)
        if res == 4:
            x = p[bp + 'const'].to(ws.dtype).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
            x = synthesis_layer(p, bp + 'conv1.', x, next(it), noise=nz(bp + 'conv1'), **lk)
        elif cfg.architecture == 'resnet':
            y = conv2d_layer(p, bp + 'skip.', x, up=2, gain=math.sqrt(0.5), low=low)
            x = synthesis_layer(p, bp + 'conv0.', x, next(it), up=2, noise=nz(bp + 'conv0'), **lk)
            x = synthesis_layer(p, bp + 'conv1.', x, next(it), noise=nz(bp + 'conv1'), gain=math.sqrt(0.5), **lk)
            x = _rv(y + x, low)
        else:
            x = synthesis_layer(p, bp + 'conv0.', x, next(it), up=2, noise=nz(bp + 'conv0'), **lk)
            if taps is not None:
                taps[f'b{res}.conv0'] = x
            x = synthesis_layer(p, bp + 'conv1.', x, next(it), noise=nz(bp + 'conv1'), **lk)
        x = _rg(x, low)          # (two consumers: the sum of their gradients is a 16-bit tensor)
        if taps is not None:
            taps[f'b{res}'] = x
        if img is not None:
            img = resample.upsample2d(img, _FILTER)
        if has_rgb:
            y = torgb_layer(p, bp + 'torgb.', x, next(it), cfg, kinds, fused_modconv=fused_modconv, low=low)
            img = img + y if img is not None else y
    return img


def generator(p, z, c, cfg, truncation_psi=1.0, truncation_cutoff=None, **synthesis_kwargs):
    ws = mapping(p, z, c, cfg, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
    return synthesis(p, ws, cfg, **synthesis_kwargs)


# ----------------------------------------------------------------------------
# Discriminator.

def discriminator(p, img, c, cfg):
    x = None
    arch = cfg.architecture
    cc = cfg.conv_clamp
    for res in cfg.block_resolutions:
        bp = f'b{res}.'
        low = _low_dtype(res)
        if x is None or arch == 'skip':
            y = conv2d_layer(p, bp + 'fromrgb.', _rv(img, low), activation='lrelu', conv_clamp=cc, low=low)
            x = _rv(x + y, low) if x is not None else y
            img = resample.downsample2d(img, _FILTER) if arch == 'skip' else None
        if arch == 'resnet':
            x = _rg(x, low)      # (two consumers: the sum of their gradients is a 16-bit tensor)
            y = conv2d_layer(p, bp + 'skip.', x, down=2, gain=math.sqrt(0.5), low=low)
            x = conv2d_layer(p, bp + 'conv0.', x, activation='lrelu', conv_clamp=cc, low=low)
            x = conv2d_layer(p, bp + 'conv1.', x, activation='lrelu', down=2, conv_clamp=cc, gain=math.sqrt(0.5), low=low)
            x = _rv(y + x, low)
        else:
            x = conv2d_layer(p, bp + 'conv0.', x, activation='lrelu', conv_clamp=cc, low=low)
            x = conv2d_layer(p, bp + 'conv1.', x, activation='lrelu', down=2, conv_clamp=cc, low=low)

    cmap_dim = cfg.cmap_dim if cfg.cmap_dim is not None else cfg.channels(4)
    if cfg.c_dim == 0:
        cmap_dim = 0
    cmap = None
    if cfg.c_dim > 0:
        cmap = mapping(p, None, c, cfg, prefix='mapping.', num_ws=None, z_dim=0, c_dim=cfg.c_dim,
                       num_layers=cfg.mapping_layers)
    if arch == 'skip':
        x = x + conv2d_layer(p, 'b4.fromrgb.', img, activation='lrelu')
    if cfg.mbstd_num_channels > 0:
        x = pointwise.minibatch_std(x, cfg.mbstd_group_size, cfg.mbstd_num_channels)
    x = conv2d_layer(p, 'b4.conv.', x, activation='lrelu', conv_clamp=cc)
    x = fully_connected(p, 'b4.fc.', x.flatten(1), activation='lrelu')
    x = fully_connected(p, 'b4.out.', x)
    if cmap_dim > 0:
        x = (x * cmap).sum(dim=1, keepdim=True) * (1 / math.sqrt(cmap_dim))
    return x
