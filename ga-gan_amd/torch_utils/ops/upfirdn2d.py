"""FIR resampling of image batches on the GPU: pad, up-sample, filter, down-sample.

Same public functions and signatures as the reference's
torch_utils/ops/upfirdn2d.py (`setup_filter`, `upfirdn2d`, `filter2d`, `upsample2d`,
`downsample2d`, `_parse_padding`, `_get_filter_size`), executed by the gfx950 kernel
behind `ga_gan_amd.upfirdn2d_plugin`.  The op is linear in x, and its adjoint is the
same op with up/down exchanged, the filter flipped and the padding recomputed
(reference: upfirdn2d.py:264-283), which makes gradients of every order available.
"""
import numpy as np
import torch

from ... import upfirdn2d_plugin as _plugin
from .. import misc


def _parse_scaling(scaling):
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    assert isinstance(scaling, (list, tuple)) and all(isinstance(v, int) for v in scaling)
    sx, sy = scaling
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple)) and all(isinstance(v, int) for v in padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    assert fw >= 1 and fh >= 1
    return fw, fh


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """float32 FIR filter: [taps] when separable (default for >= 8 taps), else [fh, fw]."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


class _Upfirdn2d(torch.autograd.Function):
    """cfg = (upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)."""

    @staticmethod
    def forward(ctx, x, f, cfg):
        upx, upy, downx, downy, px0, px1, py0, py1, flip, gain = cfg
        assert isinstance(x, torch.Tensor) and x.ndim == 4
        if f is None:
            f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
        assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
        if f.ndim == 2:
            y = _plugin.upfirdn2d(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain)
        else:  # separable: one horizontal and one vertical pass, sqrt(gain) each
            g = float(np.sqrt(gain))
            y = _plugin.upfirdn2d(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip, g)
            y = _plugin.upfirdn2d(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip, g)
        ctx.save_for_backward(f)
        ctx.cfg = cfg
        ctx.x_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        f, = ctx.saved_tensors
        upx, upy, downx, downy, px0, px1, py0, py1, flip, gain = ctx.cfg
        _, _, ih, iw = ctx.x_shape
        _, _, oh, ow = dy.shape
        fw, fh = _get_filter_size(f)
        dx = None
        if ctx.needs_input_grad[0]:
            adj = (downx, downy, upx, upy,
                   fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1,
                   fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1,
                   not flip, gain)
            dx = _Upfirdn2d.apply(dy, f, adj)
        assert not ctx.needs_input_grad[1]
        return dx, None, None


class _Branch(torch.autograd.Function):
    """(x, upfirdn2d(x, f, cfg)) for a tensor that has a second consumer beside the filter - the input of a residual
    discriminator block feeds `skip` (a down-sampling filter in front of its 1x1 convolution) and `conv0`
    (training/networks.py:792-808).  The gradient of such a tensor is the sum of what the two consumers return;
    autograd adds them in a pass of its own (read both, write one).  Here the backward hands the gradient that
    arrived for x to the kernel that applies the filter's adjoint, which adds it while it stores (`sg_upfirdn2d_add`).
    cfg as in `_Upfirdn2d`; two-dimensional filters only."""

    @staticmethod
    def forward(ctx, x, f, cfg):
        upx, upy, downx, downy, px0, px1, py0, py1, flip, gain = cfg
        assert isinstance(f, torch.Tensor) and f.ndim == 2
        y = _plugin.upfirdn2d(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain)
        ctx.save_for_backward(f)
        ctx.cfg = cfg
        ctx.x_shape = x.shape
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, gx, gy):
        f, = ctx.saved_tensors
        if gy is None or not ctx.needs_input_grad[0]:
            return (gx if ctx.needs_input_grad[0] else None), None, None
        upx, upy, downx, downy, px0, px1, py0, py1, flip, gain = ctx.cfg
        _, _, ih, iw = ctx.x_shape
        _, _, oh, ow = gy.shape
        fw, fh = _get_filter_size(f)
        adj = (downx, downy, upx, upy,
               fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1,
               fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1,
               not flip, gain)
        if gx is not None and not torch.is_grad_enabled():   # (a higher-order graph keeps the two differentiable steps)
            both = _plugin.upfirdn2d_add(gy.contiguous(), f, *adj, addend=gx.contiguous())
            if both is not None:
                return both, None, None
        d = _Upfirdn2d.apply(gy, f, adj)
        return (d if gx is None else gx + d), None, None


def branch(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """(x, upfirdn2d(x, f, up, down, padding, flip_filter, gain)): use the first result wherever x is consumed again, so
    that the gradients of both uses meet inside the filter's backward instead of in a separate sum."""
    misc.require_gpu(x, 'upfirdn2d')
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    return _Branch.apply(x, f, (upx, upy, downx, downy, px0, px1, py0, py1, bool(flip_filter), float(gain)))


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Pad, up-sample (zero insertion), FIR-filter and down-sample x[N, C, H, W].

    Arguments as the reference's `upfirdn2d` (torch_utils/ops/upfirdn2d.py:130-174).
    `impl='ref'` does not exist here: there is no CPU fallback in the product path.
    """
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl != 'cuda':
        raise NotImplementedError("impl='ref' is not part of this build (the CPU restatement is the test oracle)")
    misc.require_gpu(x, 'upfirdn2d')
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    return _Upfirdn2d.apply(x, f, (upx, upy, downx, downy, px0, px1, py0, py1, bool(flip_filter), float(gain)))


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Filter without resampling; output padded to the input's shape."""
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Up-sample by an integer factor; output shape is a multiple of the input's."""
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Down-sample by an integer factor; output shape is a fraction of the input's."""
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
