"""Convolution with optional FIR up/down-sampling (reference:
torch_utils/ops/conv2d_resample.py:59-154).  The filter passes run on the gfx950
upfirdn2d kernel; padding is applied once, before the first operation."""
import torch

from .. import misc
from . import conv2d_gradfix
from . import upfirdn2d
from .. import memo
from .upfirdn2d import _get_filter_size, _parse_padding


ROW_ALIGN = 8   # pixels: rows of intermediates are padded to whole 16-byte packets of 16-bit elements


def _align_extra(width):
    """Zero columns to append so that `width` becomes a multiple of ROW_ALIGN."""
    return (-width) % ROW_ALIGN


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True, **extra):
    """`flip_weight=True` is correlation (what conv2d computes), False is true convolution."""
    if not flip_weight:
        w = memo.memo(w, 'flip', lambda: w.flip([2, 3]))
    op = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    return op(x, w, stride=stride, padding=padding, groups=groups, **extra)


@misc.profiled_function
def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1
    assert isinstance(down, int) and down >= 1
    assert isinstance(groups, int) and groups >= 1
    out_channels, in_channels_per_group, kh, kw = (int(s) for s in w.shape)
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)

    # Padding that the resampling filters need on top of the caller's.
    if up > 1:
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2
        py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2
        px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2
        py1 += (fh - down) // 2
    pads = [px0, px1, py0, py1]
    one_by_one = (kw == 1 and kh == 1)

    if one_by_one and down > 1 and up == 1:      # decimate first: 1x1 conv on a quarter of the pixels
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, padding=pads, flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)

    if one_by_one and up > 1 and down == 1:      # 1x1 conv first, interpolate afterwards
        x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d.upfirdn2d(x=x, f=f, up=up, padding=pads, gain=up ** 2, flip_filter=flip_filter)

    if down > 1 and up == 1:                     # blur, then strided conv
        # The blurred intermediate has an odd width (W + 1).  Blur a few extra zero-padded columns so its
        # rows are whole 16-byte packets, and let the strided conv produce only the columns it would have
        # produced anyway: same values, aligned vector loads in the convolution kernels.
        out_h = (x.shape[2] + py0 + py1 - fh + 1 - kh) // down + 1
        out_w = (x.shape[3] + px0 + px1 - fw + 1 - kw) // down + 1
        extra = _align_extra(x.shape[3] + px0 + px1 - fw + 1) if (groups == 1 and x.is_cuda) else 0
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0, px1 + extra, py0, py1], flip_filter=flip_filter)
        if extra:
            return _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight, output_size=(out_h, out_w))
        return _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight)

    if up > 1:                                   # transposed strided conv, then blur (and decimate)
        if groups == 1:
            w = memo.memo(w, 'T', lambda: w.transpose(0, 1))
        else:
            w = w.reshape(groups, out_channels // groups, in_channels_per_group, kh, kw).transpose(1, 2)
            w = w.reshape(groups * in_channels_per_group, out_channels // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        # The transposed conv's output has an odd width ((W - 1) * up + k - 2 * pxt).  Ask it for extra
        # (all-zero) columns up to whole 16-byte packets and crop them again with the blur's padding.
        natural_w = (x.shape[3] - 1) * up + kw - 2 * pxt
        extra = _align_extra(natural_w) if (groups == 1 and x.is_cuda and x.dtype != torch.float64) else 0
        x = _conv2d_wrapper(x=x, w=w, stride=up, padding=[pyt, pxt], groups=groups, transpose=True,
                            flip_weight=(not flip_weight), **(dict(output_padding=[0, extra]) if extra else {}))
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0 + pxt, px1 + pxt - extra, py0 + pyt, py1 + pyt], gain=up ** 2,
                                flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
        return x

    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:   # plain conv with symmetric padding
        return _conv2d_wrapper(x=x, w=w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)

    # Anything else: pad / interpolate, convolve, decimate.
    x = upfirdn2d.upfirdn2d(x=x, f=(f if up > 1 else None), up=up, padding=pads, gain=up ** 2, flip_filter=flip_filter)
    x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
    return x
