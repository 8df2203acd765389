"""StyleGAN2 generator / discriminator with the StyleDomain few-shot parameterisations
(Affine+, AffineLight+, StyleSpace offsets), running on the gfx950 op API.

Drop-in for the reference's `training/networks.py`: same class names, constructor
arguments, forward arguments and `state_dict()` names / shapes, so a checkpoint of
the reference loads with `load_state_dict(strict=True)`.  Differences, all
deliberate:
  * blocks flagged `use_fp16` run in `LOW_PRECISION_DTYPE` (bfloat16 by default on
    MI355X; float16 is still accepted and then gets the reference's overflow
    pre-normalisation, training/networks.py:621-627);
  * the demodulation coefficients are computed as rsqrt(s^2 @ sum_k(W^2)^T + eps)
    instead of materialising the per-sample weight tensor [N, O, I, k, k]
    (training/networks.py:633-636) - same value, 1/(9 N)-th of the memory traffic;
  * `SynthesisBlock.forward` up-samples the image skip only when one exists (the
    fork's lines 1057-1063 crash on the first block; this is the upstream wiring);
  * `MinibatchStdLayer` is one kernel.
There is no CPU path: every op refuses CPU tensors.
"""
import math
import os
import re

import numpy as np
import torch

from ..torch_utils import memo
from ..torch_utils import misc
from ..torch_utils import persistence
from ..torch_utils.ops import bias_act
from ..torch_utils.ops import conv2d_resample
from ..torch_utils.ops import fma
from ..torch_utils.ops import styles as styles_op
from ..torch_utils.ops import fused_layer
from ..torch_utils.ops import minibatch_std
from ..torch_utils.ops import upfirdn2d

# dtype of the blocks the reference runs in float16 (`use_fp16`).
LOW_PRECISION_DTYPE = torch.bfloat16


# Run convolution layers as fused kernels (contraction + demodulation + noise + bias + activation in
# one pass) where the geometry allows; False = the reference's op-by-op composition.
FUSE_LAYERS = True


def set_low_precision_dtype(dtype):
    global LOW_PRECISION_DTYPE
    assert dtype in (torch.bfloat16, torch.float16, torch.float32)
    LOW_PRECISION_DTYPE = dtype


# ----------------------------------------------------------------------------
# Domain-modulation parameterisations (reference: training/networks.py:24-53, 87-137).

_RANKS = [1, 5, 10, 20, 50, 100, 256, 512]
_BASE = ['in', 'out', 'spatial', 'in_spatial', 'out_spatial', 'out_in']
out_in_base_parametrizations = _BASE + [b + '_additive' for b in _BASE]
_DECOMP = (['out+in'] + [f'out_in_{k}' for k in _RANKS] + [f'out_in_{k}_dual' for k in _RANKS]
           + [f'out_in_{k}_{t}{suffix}' for suffix in ('', '_train_in', '_train_out')
              for k in _RANKS for t in range(1, 512 // k + 1)])
out_in_decomposition_parametrizations = _DECOMP + [d + '_additive' for d in _DECOMP]
_AFFINE = [f'affine_out_in_{k}_{t}' for k in _RANKS for t in range(1, 512 // k + 1)]
affine_out_in_decomposition_parametrizations = _AFFINE + [a + '_additive' for a in _AFFINE]

_WEIGHT_KINDS = set(out_in_base_parametrizations) | set(out_in_decomposition_parametrizations)
_AFFINE_KINDS = set(affine_out_in_decomposition_parametrizations)
_STYLE_KINDS = ('multiplicative', 'additive', 'multiplicative_w_space', 'additive_w_space')
_DECOMP_RE = re.compile(r'^(?:affine_)?out_in_([0-9]+)(?:_(dual|[0-9]+))?(?:_(train_in|train_out))?(?:_additive)?$')


def split_parameterization(use_domain_modulation, domain_modulation_parametrization):
    """'a,b,c' -> (S/W-space kind, conv-weight kind, affine-weight kind); at most one of each."""
    if not use_domain_modulation:
        return None, None, None
    names = domain_modulation_parametrization.split(',')
    groups = ([n for n in names if n not in _WEIGHT_KINDS and n not in _AFFINE_KINDS],
              [n for n in names if n in _WEIGHT_KINDS],
              [n for n in names if n in _AFFINE_KINDS])
    for g in groups:
        if len(g) > 1:
            raise ValueError(f'Too many types of domain modulation: {g}')
    return tuple(g[0] if g else None for g in groups)


def _unit_rows(t):
    if torch.linalg.norm(t).item() > 1e-6:
        t /= torch.linalg.norm(t, dim=1, keepdim=True)


def _unit_cols(t):
    if torch.linalg.norm(t).item() > 1e-6:
        t /= torch.linalg.norm(t, dim=0, keepdim=True)


def _register_low_rank_terms(module, kind, out_dim, in_dim):
    """Low-rank offset terms  sum_t out_t[O, K] @ in_t[K, I].  Which factor is trained
    (Parameter) and which is a fixed random projection (buffer) depends on the kind
    (reference: training/networks.py:233-410 and 163-216)."""
    ins, outs = [], []
    if kind.startswith('out+in'):
        module.register_parameter('weights_offset_in_0', torch.nn.Parameter(torch.zeros(1, in_dim)))
        module.register_parameter('weights_offset_out_0', torch.nn.Parameter(torch.zeros(out_dim, 1)))
        ins, outs = ['weights_offset_in_0'], ['weights_offset_out_0']
    else:
        m = _DECOMP_RE.match(kind)
        if m is None:
            raise ValueError(f'Unknown type of domain modulation: {kind}')
        rank, second, train = int(m.group(1)), m.group(2), m.group(3)

        def param(name, tensor):
            module.register_parameter(name, torch.nn.Parameter(tensor))

        def buf(name, tensor):
            module.register_buffer(name, tensor)

        if second == 'dual':   # A1 @ B1 + A2 @ B2 with B1 and A2 fixed
            buf('weights_offset_in_0', torch.randn(rank, in_dim))
            param('weights_offset_in_1', torch.zeros(rank, in_dim))
            param('weights_offset_out_0', torch.zeros(out_dim, rank))
            buf('weights_offset_out_1', torch.randn(out_dim, rank))
            ins, outs = ['weights_offset_in_0', 'weights_offset_in_1'], ['weights_offset_out_0', 'weights_offset_out_1']
        else:
            terms = int(second) if second is not None else 1
            for t in range(terms):
                if second is None:                      # single term, both factors trained from zero
                    param(f'weights_offset_in_{t}', torch.zeros(rank, in_dim))
                    param(f'weights_offset_out_{t}', torch.zeros(out_dim, rank))
                elif train is None:                     # both trained, `in` starts random
                    param(f'weights_offset_in_{t}', torch.randn(rank, in_dim))
                    param(f'weights_offset_out_{t}', torch.zeros(out_dim, rank))
                elif train == 'train_in':
                    param(f'weights_offset_in_{t}', torch.zeros(rank, in_dim))
                    buf(f'weights_offset_out_{t}', torch.randn(out_dim, rank))
                else:                                   # train_out
                    buf(f'weights_offset_in_{t}', torch.randn(rank, in_dim))
                    param(f'weights_offset_out_{t}', torch.zeros(out_dim, rank))
                ins.append(f'weights_offset_in_{t}')
                outs.append(f'weights_offset_out_{t}')
    with torch.no_grad():
        for name in ins:
            _unit_rows(getattr(module, name).data)
        for name in outs:
            _unit_cols(getattr(module, name).data)
    module.weights_offset_in, module.weights_offset_out = ins, outs


def register_affine_weights_domain_modulation(self):
    kind = self.affine_weights_domain_modulation_parametrization
    if kind not in _AFFINE_KINDS:
        raise ValueError(f'Unknown type of domain modulation: {kind}')
    out_features, in_features = self.weight.shape
    _register_low_rank_terms(self, kind, out_features, in_features)
    self.register_buffer('ones_weights', self.weight.new_ones(out_features, in_features))


def register_weights_domain_modulation(self):
    kind = self.weights_domain_modulation_parametrization
    out_channels, in_channels, k0, k1 = self.weight.shape
    base = kind[:-len('_additive')] if kind.endswith('_additive') else kind
    base_shapes = {'in': (1, in_channels, 1, 1), 'out': (out_channels, 1, 1, 1), 'spatial': (1, 1, k0, k1),
                   'in_spatial': (1, in_channels, k0, k1), 'out_spatial': (out_channels, 1, k0, k1),
                   'out_in': (out_channels, in_channels, 1, 1)}
    if base in base_shapes:
        self.weights_offset = torch.nn.Parameter(torch.zeros(*base_shapes[base]))
        self.register_buffer('ones_weights', torch.ones_like(self.weights_offset))
    elif kind in out_in_decomposition_parametrizations:
        _register_low_rank_terms(self, kind, out_channels, in_channels)
        self.register_buffer('ones_weights', self.weight.new_ones(out_channels, in_channels, 1, 1))
    else:
        raise ValueError(f'Unknown type of domain modulation: {kind}')


def register_domain_modulation(self, in_channels, w_dim):
    kind = self.domain_modulation_parametrization
    if kind not in _STYLE_KINDS:
        raise ValueError(f'Unknown type of domain modulation: {kind}')
    dim = w_dim if kind.endswith('w_space') else in_channels
    if kind.startswith('multiplicative'):
        self.register_buffer('ones', torch.ones((1, dim)))
    self.offset = torch.nn.Parameter(torch.zeros(1, dim))
    self.register_buffer('offset_mask', torch.ones_like(self.offset))


def register_modulation(self, in_channels, w_dim, use_domain_modulation, domain_modulation_parametrization,
                        generator_requires_grad_parts):
    """Create the offset parameters a layer needs for the chosen parameterisation.  Conv-weight
    offsets exist only on layers that `generator_requires_grad_parts` selects
    ('all', '<synt|tRGB>_weights_offset' or '<...>_weights_offset.b<res>')."""
    self.use_domain_modulation = use_domain_modulation
    (self.domain_modulation_parametrization, self.weights_domain_modulation_parametrization,
     self.affine.affine_weights_domain_modulation_parametrization) = split_parameterization(
        use_domain_modulation, domain_modulation_parametrization)
    if self.domain_modulation_parametrization is not None:
        register_domain_modulation(self, in_channels, w_dim)
    group = 'tRGB_weights_offset' if self.is_rgb else 'synt_weights_offset'
    selected = any(p in generator_requires_grad_parts for p in ('all', group, f'{group}.b{self.resolution:d}'))
    if selected and self.weights_domain_modulation_parametrization is not None:
        register_weights_domain_modulation(self)
    else:
        self.weights_domain_modulation_parametrization = None
    if self.affine.affine_weights_domain_modulation_parametrization is not None:
        register_affine_weights_domain_modulation(self.affine)


def get_style_space_modifications(layer_idx, styles, style_space_modifications):
    """Per-layer edit vector and direction mask from ((layer, channel), magnitude, factor) triples."""
    if style_space_modifications is None:
        return None, None
    shape = [1] * styles.ndim
    shape[1] = styles.shape[1]
    modifications = torch.zeros(shape, device=styles.device, dtype=styles.dtype)
    offset_factors = torch.ones_like(modifications)
    for (layer, channel), magnitude, offset_factor in style_space_modifications:
        if layer_idx == layer:
            modifications[:, channel] = magnitude
            offset_factors[:, channel] = offset_factor
    return modifications, offset_factors


def w_to_s(self, w, weight_gain, style_space_modifications=None, style_space_modifications_first=False,
           saved_styles=None):
    """W -> S: optional W-space offset, affine layer, optional S-space offset and edits."""
    kind = self.domain_modulation_parametrization if self.use_domain_modulation else None
    if kind == 'multiplicative_w_space':
        w = (self.ones + self.offset * self.offset_mask) * w
    elif kind == 'additive_w_space':
        w = w + self.offset * self.offset_mask

    styles = self.affine(w)
    if weight_gain != 1:
        styles = styles * weight_gain
    if saved_styles is not None:
        saved_styles['initial'][self.layer_idx] = styles.detach().cpu()

    modifications, offset_factors = get_style_space_modifications(self.layer_idx, styles, style_space_modifications)
    if style_space_modifications is not None and style_space_modifications_first:
        styles = styles + modifications
    if kind in ('multiplicative', 'additive'):
        mask = self.offset_mask if style_space_modifications is None else self.offset_mask * offset_factors
        styles = (self.ones + self.offset * mask) * styles if kind == 'multiplicative' else styles + self.offset * mask
    if style_space_modifications is not None and not style_space_modifications_first:
        styles = styles + modifications
    if saved_styles is not None:
        saved_styles['final'][self.layer_idx] = styles.detach().cpu()
    return styles


def weight_to_weight(self, weight):
    """Weight with its domain offset applied: W + dW ('..._additive') or (1 + dW) * W."""
    affine_kind = getattr(self, 'affine_weights_domain_modulation_parametrization', None)
    conv_kind = getattr(self, 'weights_domain_modulation_parametrization', None)
    if affine_kind is not None:       # low-rank terms are summed
        offset = sum(getattr(self, o) @ getattr(self, i) for o, i in zip(self.weights_offset_out, self.weights_offset_in))
        return weight + offset if affine_kind.endswith('additive') else (self.ones_weights + offset) * weight
    if conv_kind is not None:
        if conv_kind in out_in_decomposition_parametrizations:
            if conv_kind.startswith('out+in'):
                offset = getattr(self, self.weights_offset_out[0]) + getattr(self, self.weights_offset_in[0])
            else:                     # low-rank terms are averaged
                offset = sum(getattr(self, o) @ getattr(self, i)
                             for o, i in zip(self.weights_offset_out, self.weights_offset_in)) / len(self.weights_offset_out)
            offset = offset.unsqueeze(2).unsqueeze(3)
        else:
            offset = self.weights_offset
        return weight + offset if conv_kind.endswith('additive') else (self.ones_weights + offset) * weight
    return weight


def weight_sources(self):
    """The tensors `weight_to_weight(self, self.weight)` reads."""
    sources = [self.weight]
    affine_kind = getattr(self, 'affine_weights_domain_modulation_parametrization', None)
    conv_kind = getattr(self, 'weights_domain_modulation_parametrization', None)
    if affine_kind is not None or conv_kind in out_in_decomposition_parametrizations:
        sources += [getattr(self, name) for name in list(self.weights_offset_out) + list(self.weights_offset_in)]
    elif conv_kind is not None:
        sources.append(self.weights_offset)
    if affine_kind is not None or conv_kind is not None:
        sources.append(self.ones_weights)
    return sources


# ----------------------------------------------------------------------------

@misc.profiled_function
def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


@misc.profiled_function
def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True, dcoefs=None):
    """Weight-modulated (and demodulated) convolution; arguments as the reference's
    `modulated_conv2d` (training/networks.py:591-668).

    x [N, I, H, W], weight [O, I, kh, kw], styles [N, I], noise broadcastable to the output.
    fused_modconv=False scales the activations before and after one shared-weight
    convolution (training); True runs one grouped convolution with per-sample weights.
    """
    batch_size = x.shape[0]
    out_channels, in_channels, kh, kw = weight.shape
    misc.assert_shape(weight, [out_channels, in_channels, kh, kw])
    misc.assert_shape(x, [batch_size, in_channels, None, None])
    misc.assert_shape(styles, [batch_size, in_channels])

    if x.dtype == torch.float16 and demodulate:   # keep float16 products in range
        weight = weight * (1 / np.sqrt(in_channels * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)

    if not demodulate:
        dcoefs = None
    elif dcoefs is None or x.dtype == torch.float16:   # rsqrt(sum_{i,k} (W[o,i,k] s[n,i])^2 + eps) without the [N,O,I,k,k] intermediate
        dcoefs = _demodulation(styles, weight)

    if not fused_modconv:
        x = x * styles.to(x.dtype).reshape(batch_size, -1, 1, 1)
        x = conv2d_resample.conv2d_resample(x=x, w=_cast(weight, x.dtype), f=resample_filter, up=up, down=down,
                                            padding=padding, flip_weight=flip_weight)
        if demodulate and noise is not None:
            x = fma.fma(x, dcoefs.to(x.dtype).reshape(batch_size, -1, 1, 1), noise.to(x.dtype))
        elif demodulate:
            x = x * dcoefs.to(x.dtype).reshape(batch_size, -1, 1, 1)
        elif noise is not None:
            x = x.add_(noise.to(x.dtype))
        return x

    w = weight.unsqueeze(0) * styles.reshape(batch_size, 1, -1, 1, 1)
    if demodulate:
        w = w * dcoefs.reshape(batch_size, -1, 1, 1, 1)
    x = x.reshape(1, -1, *x.shape[2:])
    w = w.reshape(-1, in_channels, kh, kw)
    x = conv2d_resample.conv2d_resample(x=x, w=w.to(x.dtype), f=resample_filter, up=up, down=down, padding=padding,
                                        groups=batch_size, flip_weight=flip_weight)
    x = x.reshape(batch_size, -1, *x.shape[2:])
    if noise is not None:
        x = x.add_(noise)
    return x


class _ScaledCast(torch.autograd.Function):
    """(tensor * gain).to(dtype) with the value taken from the owner's cache of derived tensors, so that a
    trainable layer evaluated several times between two optimizer steps (the discriminator in Dmain) prepares and
    packs its weight once; the gradient is gain * g in the parameter's dtype."""

    @staticmethod
    def forward(ctx, tensor, owner, key, gain, dtype):
        ctx.gain, ctx.src_dtype = gain, tensor.dtype
        base = memo.derived(owner, key, [tensor], lambda: _scaled_cast(tensor, gain, dtype), values_only=True)
        return memo.share(base.view_as(base), base)

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled():      # a higher-order graph is wanted: differentiable composition
            g = g.to(ctx.src_dtype)
            return (g * ctx.gain if ctx.gain != 1 else g), None, None, None, None
        return _scaled_cast(g, ctx.gain, ctx.src_dtype), None, None, None, None


_GAINS = {}


def _gain_vector(gain, device):
    """float32 [1] tensor holding `gain` (a one-element vector, unlike a scalar, takes part in type promotion)."""
    key = (float(gain), device)
    vec = _GAINS.get(key)
    if vec is None:
        vec = torch.full([1], float(gain), dtype=torch.float32, device=device)
        if not (vec.is_cuda and torch.cuda.is_current_stream_capturing()):
            _GAINS[key] = vec
    return vec


def _scaled_cast(tensor, gain, dtype):
    """(tensor * gain).to(dtype) without a graph, as one kernel: the product is formed in float32 and rounded once."""
    if gain == 1 or tensor.dtype == dtype:
        return tensor.to(dtype) if gain == 1 else tensor * gain
    if dtype == torch.float32:           # 16-bit -> float32: promotion against the float32 vector widens first
        return tensor * _gain_vector(gain, tensor.device)
    if tensor.dtype == torch.float32:    # float32 -> 16-bit: product in float32, written in the narrow type
        return torch.mul(tensor, gain, out=torch.empty_like(tensor, dtype=dtype))
    return (tensor * gain).to(dtype)


def _prepared(owner, tensor, gain, dtype, key):
    """(tensor * gain).to(dtype) for a parameter (or buffer) of the module `owner`, once per parameter version."""
    if torch.is_grad_enabled() and tensor.requires_grad:
        return _ScaledCast.apply(tensor, owner, (key, dtype), gain, dtype)
    return memo.derived(owner, (key, dtype), [tensor], lambda: _scaled_cast(tensor, gain, dtype))


_EPS = {}


def _demodulation(styles, weight):
    """rsqrt(sum_{i,k} (W[o,i,k] s[n,i])^2 + 1e-8) as one matrix product of s^2 [N, I] with sum_k W^2 [I, O]
    (kept per weight version), the epsilon folded into the product's additive term."""
    wsq_t = memo.memo(weight, 'wsq^T', lambda: weight.square().sum(dim=[2, 3]).t())
    key = (styles.device, styles.dtype)
    eps = _EPS.get(key)
    if eps is None:
        eps = torch.full([1], 1e-8, device=styles.device, dtype=styles.dtype)
        if not (styles.is_cuda and torch.cuda.is_current_stream_capturing()):
            _EPS[key] = eps
    return torch.addmm(eps, styles.square(), wsq_t.to(styles.dtype)).rsqrt()


def _cast(weight, dtype):
    return memo.memo(weight, ('cast', dtype), lambda: weight.to(dtype))


FUSE_STYLES = os.environ.get('SG_FUSE_STYLES', '1') != '0'    # (off: the framework's operators, for timing the difference)


def _layer_styles(layer, w, weight_gain, demod_weight, kwargs):
    """(styles, demodulation coefficients or None) of a modulated layer from its row `w` of the latent codes.

    A layer without style-space offsets or edits gets both from one operator (torch_utils/ops/styles.py) that reads
    the affine layer's parameters as they are stored; `demod_weight` is the convolution weight whose squared sums the
    coefficients need (None: the caller does not demodulate, or computes the coefficients itself).  Every other case
    goes through w_to_s, and the caller derives the coefficients."""
    plain = (FUSE_STYLES and not kwargs and w.is_cuda and w.dtype == torch.float32
             and (not layer.use_domain_modulation or layer.domain_modulation_parametrization is None))
    if not plain:
        return w_to_s(layer, w=w, weight_gain=weight_gain, **kwargs), None
    affine = layer.affine
    if getattr(affine, 'affine_weights_domain_modulation_parametrization', None) is None:
        a, a_gain = affine.weight, affine.weight_gain
    else:
        a, a_gain = memo.derived(affine, ('w', w.dtype), weight_sources(affine),
                                 lambda: weight_to_weight(affine, affine.weight.to(w.dtype)) * affine.weight_gain), 1.0
    wsq_t = None
    if demod_weight is not None:
        wsq_t = memo.memo(demod_weight, 'wsq^T dense', lambda: demod_weight.square().sum(dim=[2, 3]).t().contiguous())
    return styles_op.styles(w, a, affine.bias, wsq_t, a_gain=a_gain, b_gain=affine.bias_gain, gain=weight_gain)


def _modulated_layer(x, weight, styles, noise, noise_strength, bias, up, padding, resample_filter, demodulate,
                     flip_weight, act, gain, clamp, dcoefs=None, also_input=False):
    """Training-path modulated convolution + bias/activation as one fused layer (same arithmetic as
    `modulated_conv2d(..., fused_modconv=False)` followed by `bias_act`, without the intermediates)."""
    out_channels, in_channels, kh, kw = weight.shape
    if x.dtype == torch.float16 and demodulate:   # keep float16 products in range
        weight = weight * (1 / np.sqrt(in_channels * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)
    if demodulate and (dcoefs is None or x.dtype == torch.float16):
        dcoefs = _demodulation(styles, weight)
    return fused_layer.conv_layer(x, _cast(weight, x.dtype), f=resample_filter, up=up, padding=padding, flip_weight=flip_weight,
                                  styles=styles, dcoefs=dcoefs, noise=noise, noise_strength=noise_strength,
                                  bias=bias, act=act, gain=gain, clamp=clamp, also_input=also_input)


# ----------------------------------------------------------------------------

@persistence.records_init
class FullyConnectedLayer(torch.nn.Module):
    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        if getattr(self, 'affine_weights_domain_modulation_parametrization', None) is None:
            w = _prepared(self, self.weight, self.weight_gain, x.dtype, 'w')
        else:
            w = memo.derived(self, ('w', x.dtype), weight_sources(self),
                             lambda: weight_to_weight(self, self.weight.to(x.dtype)) * self.weight_gain)
        b = _prepared(self, self.bias, self.bias_gain, x.dtype, 'b') if self.bias is not None else None
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)


@persistence.records_init
class Conv2dLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        self.activation = activation
        self.up = up
        self.down = down
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        memory_format = torch.channels_last if channels_last else torch.contiguous_format
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format)
        bias = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(bias) if bias is not None else None
        else:   # Freeze-D: same names, but buffers
            self.register_buffer('weight', weight)
            if bias is not None:
                self.register_buffer('bias', bias)
            else:
                self.bias = None

    def forward(self, x, gain=1, also_input=False, residual=None):
        """`also_input=True` returns (x, y) for a caller that feeds x to a second layer as well: it must use the returned
        x there (a 16-bit down-sampling 1x1 layer then receives both gradients of x in its filter's backward kernel).
        `residual`: a callable; the result is layer(x) + residual(x) - the two branches of a residual block, with the
        other branch evaluated first so that this layer's kernel can add it while it stores its own result."""
        if residual is not None:
            return self._with_residual(x, gain, residual)
        w = _prepared(self, self.weight, self.weight_gain, x.dtype, 'w')
        b = _prepared(self, self.bias, 1, x.dtype, 'b') if self.bias is not None else None
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        if FUSE_LAYERS and self.activation in ('linear', 'lrelu') and fused_layer.supported(x, w, self.up, self.down):
            if also_input and self.down == 2 and w.shape[2] == 1 and x.dtype != torch.float32 and x.requires_grad \
                    and x.is_contiguous() and self.resample_filter.ndim == 2:
                x, xd = fused_layer.branch_down(x, self.resample_filter, self.padding)
                return x, fused_layer.conv_layer(xd, w, f=self.resample_filter, up=1, down=2, padding=self.padding, bias=b,
                                                 act=self.activation, gain=self.act_gain * gain, clamp=act_clamp,
                                                 prefiltered=True)
            y = fused_layer.conv_layer(x, w, f=self.resample_filter, up=self.up, down=self.down, padding=self.padding,
                                       flip_weight=(self.up == 1), bias=b, act=self.activation,
                                       gain=self.act_gain * gain, clamp=act_clamp)
            return (x, y) if also_input else y
        y = conv2d_resample.conv2d_resample(x=x, w=w, f=self.resample_filter, up=self.up, down=self.down,
                                            padding=self.padding, flip_weight=(self.up == 1))
        y = bias_act.bias_act(y, b, act=self.activation, gain=self.act_gain * gain, clamp=act_clamp)
        return (x, y) if also_input else y


def _conv2d_layer_with_residual(self, x, gain, residual):
    w = _prepared(self, self.weight, self.weight_gain, x.dtype, 'w')
    b = _prepared(self, self.bias, 1, x.dtype, 'b') if self.bias is not None else None
    act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
    if not (FUSE_LAYERS and self.activation in ('linear', 'lrelu') and fused_layer.supported(x, w, self.up, self.down)):
        x, y = self.forward(x, gain=gain, also_input=True)
        return y.add_(residual(x))
    kwargs = dict(f=self.resample_filter, up=self.up, down=self.down, padding=self.padding, flip_weight=(self.up == 1), bias=b,
                  act=self.activation, gain=self.act_gain * gain, clamp=act_clamp)
    if self.down == 2 and w.shape[2] == 1 and x.dtype != torch.float32 and x.requires_grad and x.is_contiguous() \
            and self.resample_filter.ndim == 2:
        x, xd = fused_layer.branch_down(x, self.resample_filter, self.padding)
        return fused_layer.conv_layer(xd, w, prefiltered=True, addend=residual(x), **kwargs)
    return fused_layer.conv_layer(x, w, addend=residual(x), **kwargs)


Conv2dLayer._with_residual = _conv2d_layer_with_residual


@persistence.records_init
class MappingNetwork(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.995):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws = z_dim, c_dim, w_dim, num_ws
        self.num_layers = num_layers
        self.w_avg_beta = w_avg_beta
        if embed_features is None:
            embed_features = w_dim
        if c_dim == 0:
            embed_features = 0
        if layer_features is None:
            layer_features = w_dim
        features = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(features[idx], features[idx + 1], activation=activation,
                                                          lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        x = None
        with torch.autograd.profiler.record_function('input'):
            if self.z_dim > 0:
                misc.assert_shape(z, [None, self.z_dim])
                x = normalize_2nd_moment(z.to(torch.float32))
            if self.c_dim > 0:
                misc.assert_shape(c, [None, self.c_dim])
                y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
                x = torch.cat([x, y], dim=1) if x is not None else y
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if self.w_avg_beta is not None and self.training and not skip_w_avg_update:
            with torch.autograd.profiler.record_function('update_w_avg'):
                self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            with torch.autograd.profiler.record_function('broadcast'):
                x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            with torch.autograd.profiler.record_function('truncate'):
                assert self.w_avg_beta is not None
                if self.num_ws is None or truncation_cutoff is None:
                    x = self.w_avg.lerp(x, truncation_psi)
                else:
                    x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


@persistence.records_init
class SynthesisLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True,
                 activation='lrelu', resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False,
                 use_domain_modulation=False, domain_modulation_parametrization='multiplicative',
                 generator_requires_grad_parts=('all',)):
        super().__init__()
        self.is_rgb = False
        self.resolution = resolution
        self.up = up
        self.use_noise = use_noise
        self.activation = activation
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        memory_format = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(
            torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        register_modulation(self, in_channels, w_dim, use_domain_modulation, domain_modulation_parametrization,
                            generator_requires_grad_parts)
        self.layer_idx = None

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1, **kwargs):
        assert noise_mode in ['random', 'const', 'none']
        in_resolution = self.resolution // self.up
        misc.assert_shape(x, [None, self.weight.shape[1], in_resolution, in_resolution])
        weight = memo.derived(self, 'weight', weight_sources(self), lambda: weight_to_weight(self, self.weight))
        styles, dcoefs = _layer_styles(self, w, 1.0, weight if x.dtype != torch.float16 else None, kwargs)
        bias = memo.derived(self, ('b', x.dtype), [self.bias], lambda: self.bias.to(x.dtype))
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        if (FUSE_LAYERS and not fused_modconv and self.activation in ('linear', 'lrelu')
                and fused_layer.supported(x, _cast(weight, x.dtype), self.up, 1)):
            unit = None
            if self.use_noise and noise_mode == 'random':
                unit = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device)
            if self.use_noise and noise_mode == 'const':
                unit = self.noise_const
            return _modulated_layer(x, weight, styles, unit, self.noise_strength if unit is not None else None,
                                    bias, self.up, self.padding, self.resample_filter, True,
                                    (self.up == 1), self.activation, self.act_gain * gain, act_clamp, dcoefs=dcoefs)
        noise = None
        if self.use_noise and noise_mode == 'random':
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device) * self.noise_strength
        if self.use_noise and noise_mode == 'const':
            noise = self.noise_const * self.noise_strength
        x = modulated_conv2d(x=x, weight=weight, styles=styles, noise=noise, up=self.up, padding=self.padding,
                             resample_filter=self.resample_filter, flip_weight=(self.up == 1),
                             fused_modconv=fused_modconv, dcoefs=dcoefs)
        return bias_act.bias_act(x, bias, act=self.activation, gain=self.act_gain * gain, clamp=act_clamp)


@persistence.records_init
class ToRGBLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=1, conv_clamp=None,
                 channels_last=False, use_domain_modulation=False, domain_modulation_parametrization='multiplicative',
                 generator_requires_grad_parts=('all',), activation=None, use_noise=None):
        super().__init__()   # activation / use_noise are accepted and ignored (legacy checkpoints pass them)
        self.is_rgb = True
        self.resolution = resolution
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        memory_format = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(
            torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        register_modulation(self, in_channels, w_dim, use_domain_modulation, domain_modulation_parametrization,
                            generator_requires_grad_parts)
        self.layer_idx = None

    def forward(self, x, w, fused_modconv=True, noise_mode='random', also_input=False, **kwargs):
        """`also_input`: returns (x, rgb); a caller that consumes x again (the next block) takes it from there, so that
        the two gradients of x meet inside this layer's backward instead of in a sum of their own."""
        styles, _ = _layer_styles(self, w, self.weight_gain, None, kwargs)
        weight = memo.derived(self, 'weight', weight_sources(self), lambda: weight_to_weight(self, self.weight))
        bias = memo.derived(self, ('b', x.dtype), [self.bias], lambda: self.bias.to(x.dtype))
        if FUSE_LAYERS and not fused_modconv and fused_layer.supported(x, _cast(weight, x.dtype), 1, 1):
            return _modulated_layer(x, weight, styles, None, None, bias, 1, 0, None, False, True,
                                    'linear', 1.0, self.conv_clamp, also_input=also_input)
        y = modulated_conv2d(x=x, weight=weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        y = bias_act.bias_act(y, bias, clamp=self.conv_clamp)
        return (x, y) if also_input else y


def _channels_per_resolution(resolutions, channel_base, channel_max):
    """Feature maps at each resolution: channel_base / resolution, capped (reference: training/networks.py:1098, :1394)."""
    return {res: min(channel_base // res, channel_max) for res in resolutions}


def _first_low_precision_resolution(img_resolution_log2, num_fp16_res):
    """The `num_fp16_res` highest resolutions run in 16 bits, never below 8 x 8 (reference: :1099, :1395)."""
    return max(2 ** (img_resolution_log2 + 1 - num_fp16_res), 8)


def _block_precision(block, force_fp32):
    """(dtype, memory format) of a block's activations."""
    low = block.use_fp16 and not force_fp32
    return (LOW_PRECISION_DTYPE if low else torch.float32,
            torch.channels_last if block.channels_last and not force_fp32 else torch.contiguous_format)


@persistence.records_init
class SynthesisBlock(torch.nn.Module):
    """One resolution of the generator: [const | conv0 (up 2)] -> conv1 -> to-RGB added to the up-sampled image.
    Sub-module and parameter names are the reference's (training/networks.py:968-1026); the forward pass is the
    upstream wiring (:1028-1074 with the image up-sampling inside `if img is not None`)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture='skip',
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, use_fp16=False, fp16_channels_last=False,
                 **layer_kwargs):
        if architecture not in ('orig', 'skip', 'resnet'):
            raise AssertionError(f'unknown architecture {architecture!r}')
        super().__init__()
        self.in_channels, self.w_dim, self.resolution = in_channels, w_dim, resolution
        self.img_channels, self.is_last, self.architecture = img_channels, is_last, architecture
        self.use_fp16 = use_fp16
        self.channels_last = use_fp16 and fp16_channels_last
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        first = in_channels == 0                      # the 4 x 4 block starts from a learned constant
        shared = dict(w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp, channels_last=self.channels_last, **layer_kwargs)
        if first:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, up=2, resample_filter=resample_filter, **shared)
        self.conv1 = SynthesisLayer(out_channels, out_channels, **shared)
        self.num_conv = 1 if first else 2
        self.num_torgb = 0
        if is_last or architecture == 'skip':
            self.torgb = ToRGBLayer(out_channels, img_channels, **shared)
            self.num_torgb = 1
        if not first and architecture == 'resnet':
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2,
                                    resample_filter=resample_filter, channels_last=self.channels_last)

    def _features(self, x, ws, layer_kwargs):
        """The block's feature maps from the previous block's (or the constant); consumes num_conv rows of ws."""
        if self.in_channels == 0:
            return self.conv1(x, ws[0], **layer_kwargs)
        if self.architecture != 'resnet':
            return self.conv1(self.conv0(x, ws[0], **layer_kwargs), ws[1], **layer_kwargs)
        shortcut = self.skip(x, gain=np.sqrt(0.5))
        x = self.conv0(x, ws[0], **layer_kwargs)
        x = self.conv1(x, ws[1], gain=np.sqrt(0.5), **layer_kwargs)
        return shortcut.add_(x)

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, **layer_kwargs):
        misc.assert_shape(ws, [None, self.num_conv + self.num_torgb, self.w_dim])
        dtype, memory_format = _block_precision(self, force_fp32)
        if fused_modconv is None:     # per-sample weights only for inference, where they need no weight gradient
            fused_modconv = (not self.training) and (dtype == torch.float32 or int(ws.shape[0]) == 1)
        layer_kwargs = dict(layer_kwargs, fused_modconv=fused_modconv)
        ws = ws.unbind(dim=1)

        if self.in_channels == 0:
            x = self.const.to(dtype=dtype, memory_format=memory_format).unsqueeze(0).repeat([ws[0].shape[0], 1, 1, 1])
        else:
            misc.assert_shape(x, [None, self.in_channels, self.resolution // 2, self.resolution // 2])
            x = x.to(dtype=dtype, memory_format=memory_format)
        x = self._features(x, ws, layer_kwargs)

        if img is not None:
            misc.assert_shape(img, [None, self.img_channels, self.resolution // 2, self.resolution // 2])
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        if self.num_torgb:
            # (x goes on to the next block: taken from the to-RGB layer, its two gradients meet in that layer's backward)
            x, rgb = self.torgb(x, ws[self.num_conv], also_input=True, **layer_kwargs)
            rgb = rgb.to(dtype=torch.float32, memory_format=torch.contiguous_format)
            img = rgb if img is None else img.add_(rgb)

        assert x.dtype == dtype and (img is None or img.dtype == torch.float32)
        return x, img


@persistence.records_init
class SynthesisNetwork(torch.nn.Module):
    """Blocks b4 .. b<img_resolution>; row k of `ws` feeds the k-th convolution in order, and a block's to-RGB
    layer shares the row of the next block's first convolution (reference: training/networks.py:1079-1132)."""

    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=0,
                 **block_kwargs):
        if img_resolution < 4 or img_resolution & (img_resolution - 1):
            raise AssertionError('img_resolution must be a power of two, at least 4')
        super().__init__()
        self.w_dim, self.img_resolution, self.img_channels = w_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels = _channels_per_resolution(self.block_resolutions, channel_base, channel_max)
        low_from = _first_low_precision_resolution(self.img_resolution_log2, num_fp16_res)
        self.num_ws = 0
        for res in self.block_resolutions:
            last = res == img_resolution
            block = SynthesisBlock(channels[res // 2] if res > 4 else 0, channels[res], w_dim=w_dim, resolution=res,
                                   img_channels=img_channels, is_last=last, use_fp16=(res >= low_from), **block_kwargs)
            self.num_ws += block.num_conv + (block.num_torgb if last else 0)
            setattr(self, f'b{res}', block)

    def forward(self, ws, **block_kwargs):
        with torch.autograd.profiler.record_function('split_ws'):
            misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
            ws = ws.to(torch.float32)
            blocks, rows, first = [], [], 0
            for res in self.block_resolutions:
                block = getattr(self, f'b{res}')
                blocks.append(block)
                rows.append(ws.narrow(1, first, block.num_conv + block.num_torgb))
                first += block.num_conv       # (the to-RGB row is the next block's first row)
        x = img = None
        for block, block_ws in zip(blocks, rows):
            x, img = block(x, img, block_ws, **block_kwargs)
        return img


@persistence.records_init
class Generator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, synthesis_kwargs={}):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels,
                                          **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)
        # layer numbering used by style-space edits: conv0, conv1, torgb per block, in module order
        idx = 0
        for name, module in self.named_modules():
            if re.match(r'synthesis.b[0-9]*.conv[0-1]$', name) or re.match(r'synthesis.b[0-9]*.torgb$', name):
                module.layer_idx = idx
                idx += 1

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.synthesis(ws, **synthesis_kwargs)


# ----------------------------------------------------------------------------

@persistence.records_init
class DiscriminatorBlock(torch.nn.Module):
    """One resolution of the discriminator: [from-RGB] -> conv0 -> conv1 (down 2), with a 1x1 down-sampling shortcut in
    the 'resnet' architecture.  Names and layer numbering (Freeze-D) follow training/networks.py:1176-1242."""

    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx,
                 architecture='resnet', activation='lrelu', resample_filter=[1, 3, 3, 1], conv_clamp=None,
                 use_fp16=False, fp16_channels_last=False, freeze_layers=0, **_):
        if in_channels not in (0, tmp_channels) or architecture not in ('orig', 'skip', 'resnet'):
            raise AssertionError('bad DiscriminatorBlock configuration')
        super().__init__()
        self.in_channels, self.resolution, self.img_channels = in_channels, resolution, img_channels
        self.first_layer_idx, self.architecture, self.use_fp16 = first_layer_idx, architecture, use_fp16
        self.channels_last = use_fp16 and fp16_channels_last
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_layers = 0

        def layer(cin, cout, kernel_size, **kwargs):
            """The block's next layer; the discriminator's first `freeze_layers` layers hold buffers, not parameters."""
            trainable = first_layer_idx + self.num_layers >= freeze_layers
            self.num_layers += 1
            return Conv2dLayer(cin, cout, kernel_size=kernel_size, trainable=trainable, channels_last=self.channels_last, **kwargs)

        self.reads_image = in_channels == 0 or architecture == 'skip'
        if self.reads_image:
            self.fromrgb = layer(img_channels, tmp_channels, 1, activation=activation, conv_clamp=conv_clamp)
        self.conv0 = layer(tmp_channels, tmp_channels, 3, activation=activation, conv_clamp=conv_clamp)
        self.conv1 = layer(tmp_channels, out_channels, 3, activation=activation, down=2, resample_filter=resample_filter,
                           conv_clamp=conv_clamp)
        if architecture == 'resnet':
            self.skip = layer(tmp_channels, out_channels, 1, bias=False, down=2, resample_filter=resample_filter)

    def forward(self, x, img, force_fp32=False):
        dtype, memory_format = _block_precision(self, force_fp32)
        if x is not None:
            misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
            x = x.to(dtype=dtype, memory_format=memory_format)
        if self.reads_image:
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            img = img.to(dtype=dtype, memory_format=memory_format)
            features = self.fromrgb(img)
            x = features if x is None else x + features
            img = upfirdn2d.downsample2d(img, self.resample_filter) if self.architecture == 'skip' else None
        if self.architecture == 'resnet':
            half = np.sqrt(0.5)                       # both branches scaled so that their sum keeps the variance
            x = self.skip(x, gain=half, residual=lambda t: self.conv1(self.conv0(t), gain=half))
        else:
            x = self.conv1(self.conv0(x))
        assert x.dtype == dtype
        return x, img


@persistence.records_init
class MinibatchStdLayer(torch.nn.Module):
    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size = group_size
        self.num_channels = num_channels

    def forward(self, x):
        return minibatch_std.minibatch_std(x, self.group_size, self.num_channels)


@persistence.records_init
class DiscriminatorEpilogue(torch.nn.Module):
    def __init__(self, in_channels, cmap_dim, resolution, img_channels, architecture='resnet', mbstd_group_size=4,
                 mbstd_num_channels=1, activation='lrelu', conv_clamp=None, **_):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels = in_channels
        self.cmap_dim = cmap_dim
        self.resolution = resolution
        self.img_channels = img_channels
        self.architecture = architecture
        if architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, in_channels, kernel_size=1, activation=activation)
        self.mbstd = MinibatchStdLayer(group_size=mbstd_group_size, num_channels=mbstd_num_channels) \
            if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation,
                                conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * (resolution ** 2), in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, 1 if cmap_dim == 0 else cmap_dim)

    def forward(self, x, img, cmap, force_fp32=False):
        misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
        x = x.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        if self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            x = x + self.fromrgb(img.to(dtype=torch.float32, memory_format=torch.contiguous_format))
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.conv(x)
        x = self.fc(x.flatten(1))
        x = self.out(x)
        if self.cmap_dim > 0:
            misc.assert_shape(cmap, [None, self.cmap_dim])
            x = (x * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(self.cmap_dim))
        assert x.dtype == torch.float32
        return x


@persistence.records_init
class Discriminator(torch.nn.Module):
    """Blocks b<img_resolution> .. b8, then the 4 x 4 epilogue b4 (reference: training/networks.py:1372-1437)."""

    def __init__(self, c_dim, img_resolution, img_channels, architecture='resnet', channel_base=32768,
                 channel_max=512, num_fp16_res=0, conv_clamp=None, cmap_dim=None, block_kwargs={}, mapping_kwargs={},
                 epilogue_kwargs={}):
        super().__init__()
        self.c_dim, self.img_resolution, self.img_channels = c_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, 2, -1)]
        channels = _channels_per_resolution(self.block_resolutions + [4], channel_base, channel_max)
        low_from = _first_low_precision_resolution(self.img_resolution_log2, num_fp16_res)
        cmap_dim = 0 if c_dim == 0 else (channels[4] if cmap_dim is None else cmap_dim)
        shared = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        layers_so_far = 0
        for res in self.block_resolutions:
            block = DiscriminatorBlock(channels[res] if res < img_resolution else 0, channels[res], channels[res // 2],
                                       resolution=res, first_layer_idx=layers_so_far, use_fp16=(res >= low_from),
                                       **block_kwargs, **shared)
            setattr(self, f'b{res}', block)
            layers_so_far += block.num_layers
        if c_dim > 0:
            self.mapping = MappingNetwork(z_dim=0, c_dim=c_dim, w_dim=cmap_dim, num_ws=None, w_avg_beta=None,
                                          **mapping_kwargs)
        self.b4 = DiscriminatorEpilogue(channels[4], cmap_dim=cmap_dim, resolution=4, **epilogue_kwargs, **shared)

    def largest_activation_elements(self):
        """Upper bound of the elements per sample of the largest tensor a pass makes (the first block's: its channels
        at the image resolution plus the blur's margin).  The kernels index a tensor with 32 bits, so callers that
        choose their own batch (the loss, when it puts both halves of Dmain through one pass) keep batch x this
        below 2**31."""
        largest = 0
        for res in self.block_resolutions:
            out_channels, tmp_channels = getattr(self, f'b{res}').conv1.weight.shape[:2]
            largest = max(largest, max(tmp_channels, self.img_channels) * (res + 2) ** 2, out_channels * (res // 2) ** 2)
        return largest

    def forward(self, img, c, **block_kwargs):
        x = None
        for res in self.block_resolutions:
            x, img = getattr(self, f'b{res}')(x, img, **block_kwargs)
        cmap = self.mapping(None, c) if self.c_dim > 0 else None
        return self.b4(x, img, cmap)
